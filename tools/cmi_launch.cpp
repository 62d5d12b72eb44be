// tools/cmi_launch.cpp -- starts one process per GPU of ONE node, before anything touches a GPU: the launcher of
// cusp::distributed jobs written in C++ (tools/bin/cg_bench --sharded, tests/cpp/bin/test_distributed).
//
//     cmi_launch -n 8 [--port 29510] -- tools/bin/cg_bench --sharded --grid=10000
//
// Each child gets RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT -- the variables torchrun sets, so
// `torchrun --no-python --nproc-per-node 8 tools/bin/cg_bench --sharded` starts the same job -- and keeps
// HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: what RCCL and peer mappings need on this pool).  The launcher itself makes no HIP call:
// fork + exec happen in a process that has never initialised a GPU.  It waits for every child; if one fails the others get SIGTERM
// (a rank stuck in a collective whose peer died would wait for ever) and the exit code is the first non-zero one.
#include <signal.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    int n = 1, port = 29510, cmd = -1;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--")) { cmd = i + 1; break; }
        if (!std::strcmp(argv[i], "-n") && i + 1 < argc) n = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--port") && i + 1 < argc) port = std::atoi(argv[++i]);
        else { std::fprintf(stderr, "usage: cmi_launch -n RANKS [--port P] -- program [args...]\n"); return 2; }
    }
    if (cmd < 0 || cmd >= argc || n < 1 || n > 64) { std::fprintf(stderr, "usage: cmi_launch -n RANKS [--port P] -- program [args...]\n"); return 2; }
    std::vector<pid_t> kids;
    for (int r = 0; r < n; r++) {
        const pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); for (pid_t k : kids) kill(k, SIGTERM); return 1; }
        if (pid == 0) {
            setenv("RANK", std::to_string(r).c_str(), 1);
            setenv("LOCAL_RANK", std::to_string(r).c_str(), 1);
            setenv("WORLD_SIZE", std::to_string(n).c_str(), 1);
            setenv("MASTER_ADDR", "127.0.0.1", 1);
            setenv("MASTER_PORT", std::to_string(port).c_str(), 1);
            setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
            execvp(argv[cmd], argv + cmd);
            std::perror(argv[cmd]);
            _exit(127);
        }
        kids.push_back(pid);
    }
    int rc = 0, left = n;
    while (left > 0) {
        int status = 0;
        const pid_t pid = wait(&status);
        if (pid < 0) break;
        left--;
        const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
        if (code != 0 && rc == 0) {
            rc = code;
            for (pid_t k : kids) if (k != pid) kill(k, SIGTERM);
        }
    }
    return rc;
}
