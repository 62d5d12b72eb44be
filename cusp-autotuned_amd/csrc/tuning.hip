// tuning.hip -- persisted launch-shape / kernel-variant table + built-in heuristics.
//
// Replaces the reference's run-time KTT tuner state (cusp/ktt/detail/ktt.inl:29-62: a process-wide
// singleton that re-tunes on every first call and is never persisted, :130-142 reset_tuning) and the
// hard-wired selector of the stock CUDA path (cusp/system/cuda/detail/multiply/csr_vector_spmv.h
// :225-258: threads-per-row from the integer mean row length).  Here the choice is made OFFLINE by
// tools/autotune (which validates every variant against the CPU oracle before timing it, as
// testing/ktt.cu:142-202 does with KTT's reference computation), written to a JSON file, and looked
// up at call time by (format, dtype, bucket of mean entries per row).  Nothing is compiled or
// allocated on the critical path.
#include "common.h"

#include <dlfcn.h>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

namespace cmi {

constexpr int kBuckets = 8; // mean entries/row in [2^b, 2^(b+1)), last bucket open-ended

struct Table {
    cmi_config cfg[CMI_TABLE_KEYS][2][kBuckets];
    bool valid[CMI_TABLE_KEYS][2][kBuckets];
    double mean[CMI_TABLE_KEYS][2][kBuckets]; // mean entries per row of the matrix the entry was tuned on (0: unknown)
    // HYB split rule per value type (cmi_tuning_hyb_rule); not set: the reference's constants
    bool hyb_valid[2];
    int hyb_kind[2];
    double hyb_relative_speed[2];
    long hyb_threshold[2];
    double hyb_light_speed[2]; // CMI_HYB_RULE_COST2: cost of a COO entry while the COO part is light (one-launch kernel)
    // csr_waver's shape and AUTO gates per value type (cmi_tuning_waver_rule); not set: the built-in defaults below
    bool waver_valid[2];
    cmi_waver_rule waver[2];
};
// csr_waver without a table: what round 4 measured (tools/autotune_waver.py, profiles/r04_autotune_waver.txt and the files it cites)
static cmi_waver_rule default_waver_rule(int dtype)
{
    cmi_waver_rule r;
    r.items_per_thread = 4;
    r.cap = 0;
    r.xcd_swizzle = 16;
    r.reserved = 0;
    // pieces of consecutive columns, entries on average: 2.5 until session 38 of round 4 -- then the regret table's fourth set had the copy
    // ahead at exactly 2.5 (5-point x 2 dof: -14 % f64, -22 % f32, missed by boundary rows) and at 2.0 in f32 (pairs of neighbours: -14 %;
    // f64 there +4 %), behind at 1.4-1.7 (5- / 7-point stencils)
    r.min_piece = dtype == CMI_F64 ? 2.2 : 1.9;
    r.min_entries = dtype == CMI_F64 ? 4400000 : 6400000; // (r04_autotune_waver.txt: the copy wins by 2 %+ on every measured matrix from here up)
    return r;
}
constexpr double kRefRelativeSpeed = 3.0; // reference csr_to_other.h:248-254
constexpr long kRefBreakeven = 4096;

static Table g_table;
static std::mutex g_mu;
static bool g_default_loaded = false;

static int bucket_of(double mean)
{
    if (!(mean > 1.0)) return 0;
    int b = (int)std::floor(std::log2(mean));
    return b < 0 ? 0 : (b >= kBuckets ? kBuckets - 1 : b);
}

static const char *kFormatNames[CMI_TABLE_KEYS] = {"csr", "ell", "dia", "coo", "hyb", "coo_sorted"};
static const char *kDtypeNames[2] = {"f64", "f32"};

static int round_block(int b)
{
    if (b <= 0) return 256;
    b = (b + kWave - 1) / kWave * kWave;
    return b > 1024 ? 1024 : b;
}

// Built-in heuristics: what runs before any table has been loaded.
static void heuristic(int format, int dtype, double mean, cmi_config *c)
{
    std::memset(c, 0, sizeof(*c));
    c->block_size = 256;
    switch (format) {
    case CMI_FORMAT_CSR:
        // every row length: stream the entry tile through LDS (coalesced 16-byte vector loads);
        // short rows are summed by one lane each (storage order, bit-exact), longer rows by a
        // power-of-two group of lanes (lane-strided partial sums + butterfly)
        c->kernel = CMI_CSR_STREAM;
        c->nontemporal = kPolStoreNT; // measured: nt y stores, plain loads, no XCD swizzle
        // measured on FEM-like stencil matrices (5/9/27-point, 27-point x 2/3/8 dof): one lane per row (its LDS
        // reads batched eight at a time) wins up to ~80 entries per row, then 32 lanes per row
        if (mean <= 12.0) { c->items_per_thread = 1; c->threads_per_row = 0; }
        else if (mean <= 96.0) { c->items_per_thread = 2; c->threads_per_row = 0; }
        else { c->items_per_thread = 2; c->threads_per_row = 32; }
        break;
    // measured (archive/tools/r2_probe.hip, archive/profiles/r02_probe_timing_session2.txt): nt policy, tiles dealt to the XCDs in chunks
    case CMI_FORMAT_ELL: c->kernel = CMI_ELL_ROW; c->items_per_thread = 1; c->nontemporal = 3; c->xcd_swizzle = 64; break;
    case CMI_FORMAT_DIA: c->kernel = CMI_DIA_ROW; c->items_per_thread = 2; c->block_size = 512; c->nontemporal = 3; c->xcd_swizzle = 32; break;
    case CMI_FORMAT_COO: c->kernel = CMI_COO_LANE4; c->items_per_thread = 4; break;
    case CMI_TABLE_COO_SORTED: c->kernel = CMI_COO_TILE; c->nontemporal = kPolStoreNT; c->xcd_swizzle = 32; break;
    default: break;
    }
    (void)dtype;
}

// Fill every zero ("default") field of a config so kernels see concrete numbers.
static void complete(int format, int dtype, int64_t rows, int64_t nnz, cmi_config *c)
{
    const double mean = rows > 0 ? (double)nnz / (double)rows : 0.0;
    if (c->kernel == CMI_KERNEL_AUTO) {
        cmi_config h;
        heuristic(format, dtype, mean, &h);
        c->kernel = h.kernel;
        if (!c->threads_per_row) c->threads_per_row = h.threads_per_row;
        if (!c->items_per_thread) c->items_per_thread = h.items_per_thread;
        if (!c->block_size) c->block_size = h.block_size;
    }
    c->block_size = round_block(c->block_size);
    if (format == CMI_FORMAT_CSR) {
        if (c->kernel == CMI_CSR_VECTOR) {
            int t = c->threads_per_row;
            if (t <= 0) { // reference rule (csr_vector_spmv.h:241-256) on the integer mean, up to 64
                const int64_t m = rows > 0 ? nnz / rows : 0;
                t = m <= 2 ? 2 : m <= 4 ? 4 : m <= 8 ? 8 : m <= 16 ? 16 : m <= 32 ? 32 : 64;
            }
            int p = 2;
            while (p < t && p < 64) p <<= 1;
            c->threads_per_row = p;
        }
        if (c->kernel == CMI_CSR_STREAM_WAVE) { // 64 rows per wave, as many entries per lane as the mean row (rounded up) has
            if (c->rows_per_block <= 0) c->rows_per_block = c->block_size;
            if (c->items_per_thread <= 0) {
                const int k = (int)std::ceil(mean);
                c->items_per_thread = k < 2 ? 2 : k > kWaveTileMaxK ? kWaveTileMaxK : k;
            }
        }
        if (c->kernel == CMI_CSR_STREAM_PIPE) {
            c->items_per_thread = 1;
            if (c->rows_per_block <= 0) {
                const int64_t tile = (int64_t)c->block_size * 4;
                double r = mean > 0.0 ? std::floor((double)(tile - 3) / mean) : (double)(c->block_size - 1);
                if (r > c->block_size - 1) r = c->block_size - 1; // one row pointer per lane
                if (r >= 32.0) r = std::floor(r / 16.0) * 16.0;   // whole 128-byte lines of y per tile
                if (r < 1.0) r = 1.0;
                c->rows_per_block = (int)r;
            }
            if (c->blocks_per_cu <= 0) c->blocks_per_cu = 8;
        }
        if (c->kernel == CMI_CSR_STREAM) {
            int ipt = c->items_per_thread;
            c->items_per_thread = ipt <= 1 ? 1 : ipt <= 2 ? 2 : 4;
            int tpr = c->threads_per_row <= 1 ? 1 : c->threads_per_row; // lanes per row in the LDS row-sum phase
            int p2 = 1;
            while (p2 < tpr && p2 < 64) p2 <<= 1;
            c->threads_per_row = p2 == 1 ? (c->threads_per_row == 1 ? 1 : 0) : p2; // 1 = storage order for EVERY row (asked for); 0 = long rows cooperative
            const int64_t tile = (int64_t)c->block_size * c->items_per_thread * 4;
            if (c->rows_per_block <= 0) {
                // largest row count whose entries fit one LDS pass (3 slots of alignment slack);
                // an explicit rows_per_block is left alone and validated by the launcher
                double r = mean > 0.0 ? std::floor((double)(tile - 3) / mean) : (double)c->block_size;
                const double max_rows = 4.0 * (c->block_size / p2); // a lane group sums at most 4 rows
                if (r > max_rows) r = max_rows;
                if (r >= 32.0) r = std::floor(r / 16.0) * 16.0; // whole 128-byte lines of y per tile
                if (r < 1.0) r = 1.0;
                c->rows_per_block = (int)r;
            }
        }
    } else {
        if (c->items_per_thread <= 0) c->items_per_thread = format == CMI_FORMAT_COO ? 4 : 1;
    }
}

// --- minimal JSON (flat objects inside "entries":[...]) --------------------------------------

static bool find_int(const std::string &obj, const char *key, long *out)
{
    std::string k = std::string("\"") + key + "\"";
    size_t p = obj.find(k);
    if (p == std::string::npos) return false;
    p = obj.find(':', p + k.size());
    if (p == std::string::npos) return false;
    char *end = nullptr;
    const char *s = obj.c_str() + p + 1;
    double v = std::strtod(s, &end);
    if (end == s) return false;
    *out = (long)v;
    return true;
}

static bool find_double(const std::string &obj, const char *key, double *out)
{
    std::string k = std::string("\"") + key + "\"";
    size_t p = obj.find(k);
    if (p == std::string::npos) return false;
    p = obj.find(':', p + k.size());
    if (p == std::string::npos) return false;
    char *end = nullptr;
    const char *s = obj.c_str() + p + 1;
    const double v = std::strtod(s, &end);
    if (end == s) return false;
    *out = v;
    return true;
}

static bool find_str(const std::string &obj, const char *key, std::string *out)
{
    std::string k = std::string("\"") + key + "\"";
    size_t p = obj.find(k);
    if (p == std::string::npos) return false;
    p = obj.find(':', p + k.size());
    if (p == std::string::npos) return false;
    size_t a = obj.find('"', p + 1);
    if (a == std::string::npos) return false;
    size_t b = obj.find('"', a + 1);
    if (b == std::string::npos) return false;
    *out = obj.substr(a + 1, b - a - 1);
    return true;
}

static int load_file(const char *path)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) { set_error("cmi_tuning_load: cannot open %s", path); return CMI_ERROR_IO; }
    std::string s;
    char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, n);
    std::fclose(f);
    // "hyb_rule": {"f64": {"relative_speed": r, "breakeven_threshold": n}, "f32": {...}}
    size_t h = s.find("\"hyb_rule\"");
    if (h != std::string::npos) {
        const size_t hend = s.find("\"entries\"", h); // the rule object precedes the entries (cmi_tuning_save's layout)
        for (int di = 0; di < 2; di++) {
            size_t k = s.find(std::string("\"") + kDtypeNames[di] + "\"", h);
            if (k != std::string::npos && hend != std::string::npos && k > hend) k = std::string::npos;
            const size_t a = k == std::string::npos ? k : s.find('{', k);
            const size_t b = a == std::string::npos ? a : s.find('}', a);
            if (b == std::string::npos) continue;
            const std::string obj = s.substr(a, b - a + 1);
            double rs = 0.0;
            long be = -1;
            std::string kind;
            if (find_double(obj, "relative_speed", &rs) && find_int(obj, "threshold", &be) && rs > 0.0 && be >= 0) {
                g_table.hyb_valid[di] = true;
                const bool has_kind = find_str(obj, "kind", &kind);
                g_table.hyb_kind[di] = (has_kind && kind == "cost2") ? CMI_HYB_RULE_COST2 : (has_kind && kind == "cost") ? CMI_HYB_RULE_COST : CMI_HYB_RULE_REFERENCE;
                g_table.hyb_relative_speed[di] = rs;
                g_table.hyb_threshold[di] = be;
                double light = 0.0;
                g_table.hyb_light_speed[di] = (find_double(obj, "light_speed", &light) && light > 0.0) ? light : rs;
            }
        }
    }
    // "waver_rule": {"f64": {"items_per_thread": 4, "cap": 0, "xcd_swizzle": 16, "min_piece": 2.5, "min_entries": 10000000}, "f32": {...}}
    size_t wr = s.find("\"waver_rule\"");
    if (wr != std::string::npos) {
        const size_t wend = s.find("\"entries\"", wr);
        for (int di = 0; di < 2; di++) {
            size_t k = s.find(std::string("\"") + kDtypeNames[di] + "\"", wr);
            if (k != std::string::npos && wend != std::string::npos && k > wend) k = std::string::npos;
            const size_t a = k == std::string::npos ? k : s.find('{', k);
            const size_t b = a == std::string::npos ? a : s.find('}', a);
            if (b == std::string::npos) continue;
            const std::string obj = s.substr(a, b - a + 1);
            long v = 0, cap = 0, swz = 0, me = 0;
            double mp = 0.0;
            if (find_int(obj, "items_per_thread", &v) && (v == 1 || v == 2 || v == 4) && find_double(obj, "min_piece", &mp) && mp >= 1.0 && find_int(obj, "min_entries", &me) && me >= 0) {
                cmi_waver_rule r = default_waver_rule(di);
                r.items_per_thread = (int)v;
                if (find_int(obj, "cap", &cap) && (cap == 0 || cap == 3 || cap == 4)) r.cap = (int)cap;
                if (find_int(obj, "xcd_swizzle", &swz) && swz >= 0) r.xcd_swizzle = (int)swz;
                r.min_piece = mp;
                r.min_entries = me;
                g_table.waver[di] = r;
                g_table.waver_valid[di] = true;
            }
        }
    }
    size_t p = s.find("\"entries\"");
    if (p == std::string::npos) { set_error("cmi_tuning_load: %s has no \"entries\"", path); return CMI_ERROR_IO; }
    p = s.find('[', p);
    int loaded = 0;
    while (p != std::string::npos) {
        size_t a = s.find('{', p);
        size_t close = s.find(']', p);
        if (a == std::string::npos || (close != std::string::npos && close < a)) break;
        size_t b = s.find('}', a);
        if (b == std::string::npos) break;
        std::string obj = s.substr(a, b - a + 1);
        std::string fmt, dt;
        long bucket = -1;
        if (find_str(obj, "format", &fmt) && find_str(obj, "dtype", &dt) && find_int(obj, "bucket", &bucket)) {
            int fi = -1, di = -1;
            for (int i = 0; i < CMI_TABLE_KEYS; i++) if (fmt == kFormatNames[i]) fi = i;
            for (int i = 0; i < 2; i++) if (dt == kDtypeNames[i]) di = i;
            if (fi >= 0 && di >= 0 && bucket >= 0 && bucket < kBuckets) {
                cmi_config c;
                std::memset(&c, 0, sizeof(c));
                long v;
                if (find_int(obj, "kernel", &v)) c.kernel = (int)v;
                if (find_int(obj, "block_size", &v)) c.block_size = (int)v;
                if (find_int(obj, "threads_per_row", &v)) c.threads_per_row = (int)v;
                if (find_int(obj, "rows_per_block", &v)) c.rows_per_block = (int)v;
                if (find_int(obj, "items_per_thread", &v)) c.items_per_thread = (int)v;
                if (find_int(obj, "nontemporal", &v)) c.nontemporal = (int)v;
                if (find_int(obj, "xcd_swizzle", &v)) c.xcd_swizzle = (int)v;
                if (find_int(obj, "blocks_per_cu", &v)) c.blocks_per_cu = (int)v;
                double tuned_mean = 0.0;
                if (!find_double(obj, "mean", &tuned_mean) || !(tuned_mean > 0.0)) tuned_mean = 0.0;
                g_table.cfg[fi][di][bucket] = c;
                g_table.valid[fi][di][bucket] = true;
                g_table.mean[fi][di][bucket] = tuned_mean;
                loaded++;
            }
        }
        p = b + 1;
    }
    (void)loaded;
    return CMI_SUCCESS;
}

// the table shipped next to the library: <libdir>/../tuned/gfx950.json
static std::string default_table_path()
{
    Dl_info info;
    if (dladdr((void *)&default_table_path, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t s = p.rfind('/');
        if (s != std::string::npos) return p.substr(0, s) + "/../tuned/gfx950.json";
    }
    return "tuned/gfx950.json";
}

static void ensure_default_loaded()
{
    if (g_default_loaded) return;
    g_default_loaded = true;
    const char *env = std::getenv("CMI_TUNING_TABLE");
    if (env && *env) {
        if (std::strcmp(env, "none") != 0) (void)load_file(env);
        return;
    }
    std::string p = default_table_path();
    FILE *f = std::fopen(p.c_str(), "rb");
    if (f) { std::fclose(f); (void)load_file(p.c_str()); }
}

void select_config(int format, int dtype, int64_t rows, int64_t cols, int64_t nnz, const cmi_config *user,
                   cmi_config *out)
{
    (void)cols;
    const double mean = rows > 0 ? (double)nnz / (double)rows : 0.0;
    if (user && user->kernel != CMI_KERNEL_AUTO) {
        *out = *user;
    } else {
        std::lock_guard<std::mutex> lk(g_mu);
        ensure_default_loaded();
        const int b = bucket_of(mean);
        if (format >= 0 && format < CMI_TABLE_KEYS && g_table.valid[format][dtype][b]) {
            *out = g_table.cfg[format][dtype][b];
            // A table entry was tuned at ONE mean row length of its bucket [2^b, 2^(b+1)); its rows per tile are scaled
            // to this matrix so that a tile keeps the tuned fill of the LDS pass: 176 rows of 5 fill 86 % of 1024 slots,
            // 176 rows of 7 would overflow the pass and fall off the single-pass path (25.3 vs 22.6 us on a 1.2M-row
            // unstructured FEM matrix), 64 rows of 16.5 where 64 rows of 26.6 were tuned leave half of it empty (20.1 vs
            // 17.6 us on a tetrahedral mesh; tools/unstructured_probe.py, tools/tet_mesh_probe.py).  Whole y lines kept;
            // an entry without a recorded mean is only ever shrunk to fit.
            if (format == CMI_FORMAT_CSR && out->kernel == CMI_CSR_STREAM && out->rows_per_block > 0 && mean > 0.0) {
                const int blk = round_block(out->block_size);
                const int ipt = out->items_per_thread <= 1 ? 1 : out->items_per_thread <= 2 ? 2 : 4;
                const int tpr = out->threads_per_row <= 1 ? 1 : out->threads_per_row;
                double fit = std::floor((double)((int64_t)blk * ipt * 4 - 3) / mean);
                const double max_rows = 4.0 * (blk / tpr);
                if (fit > max_rows) fit = max_rows;
                double want = (double)out->rows_per_block;
                const double tuned_mean = g_table.mean[format][dtype][b];
                const bool one_row_per_lane = out->rows_per_block <= blk / tpr; // the tuned shape's regime: keep it
                if (tuned_mean > 0.0) want = std::floor(want * tuned_mean / mean + 0.5);
                if (want >= 32.0) want = std::floor(want / 16.0 + 0.5) * 16.0; // whole y lines, nearest (keeps the tuned fill)
                if (want > fit) want = fit;
                if (one_row_per_lane && want > (double)(blk / tpr)) want = (double)(blk / tpr);
                if (want >= 32.0) want = std::floor(want / 16.0) * 16.0;
                if (want < 1.0) want = 1.0;
                out->rows_per_block = (int)want;
            }
            // The table's nt-load bit was measured on matrices far larger than the 256 MiB Infinity Cache, where the once-read
            // streams only push the vectors out of it.  A matrix whose index and value streams fit there is served from it on every
            // multiply after the first if they are loaded plainly: thermal2-like (100 MB) 20.0 us against 22.5 with the hint, a
            // 9-point matrix of 243 MB 42.2 against 48.5; at 430 MB and above the hint wins (archive/profiles/r02_stream_shape_ab.txt,
            // r02_autotune_csr_short_rows.jsonl.gz).  So the bit is dropped below 1.25x the cache.
            if (format == CMI_FORMAT_CSR && out->kernel == CMI_CSR_STREAM && (out->nontemporal & kPolLoadNT)) {
                const int64_t stream_bytes = nnz * (int64_t)(sizeof(int) + (dtype == CMI_F64 ? 8 : 4));
                if (stream_bytes <= kInfinityCacheBytes + kInfinityCacheBytes / 4) out->nontemporal &= ~kPolLoadNT;
            }
        } else
            heuristic(format, dtype, mean, out);
        if (user) { // AUTO kernel but explicit launch-shape overrides
            if (user->block_size) out->block_size = user->block_size;
            if (user->nontemporal) out->nontemporal = user->nontemporal;
        }
    }
    complete(format, dtype, rows, nnz, out);
}

} // namespace cmi

using namespace cmi;

CMI_API int cmi_tuning_load(const char *path)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_default_loaded = true;
    if (!path) path = std::getenv("CMI_TUNING_TABLE");
    if (!path || !*path) return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_load: no path and CMI_TUNING_TABLE unset");
    return load_file(path);
}

CMI_API int cmi_tuning_save(const char *path)
{
    if (!path) return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_save: null path");
    std::lock_guard<std::mutex> lk(g_mu);
    ensure_default_loaded(); // save what select_config would use, not an empty table
    FILE *f = std::fopen(path, "wb");
    if (!f) { set_error("cmi_tuning_save: cannot open %s", path); return CMI_ERROR_IO; }
    std::fprintf(f, "{\n  \"arch\": \"gfx950\",\n  \"version\": %d,\n", CMI_VERSION);
    if (g_table.hyb_valid[0] || g_table.hyb_valid[1]) {
        std::fprintf(f, "  \"hyb_rule\": {");
        bool first_rule = true;
        for (int di = 0; di < 2; di++)
            if (g_table.hyb_valid[di]) {
                std::fprintf(f, "%s\"%s\": {\"kind\": \"%s\", \"relative_speed\": %.4f, \"threshold\": %ld, \"light_speed\": %.4f}", first_rule ? "" : ", ",
                             kDtypeNames[di], g_table.hyb_kind[di] == CMI_HYB_RULE_COST2 ? "cost2" : g_table.hyb_kind[di] == CMI_HYB_RULE_COST ? "cost" : "reference",
                             g_table.hyb_relative_speed[di], g_table.hyb_threshold[di], g_table.hyb_light_speed[di]);
                first_rule = false;
            }
        std::fprintf(f, "},\n");
    }
    if (g_table.waver_valid[0] || g_table.waver_valid[1]) {
        std::fprintf(f, "  \"waver_rule\": {");
        bool first_rule = true;
        for (int di = 0; di < 2; di++)
            if (g_table.waver_valid[di]) {
                const cmi_waver_rule &r = g_table.waver[di];
                std::fprintf(f, "%s\"%s\": {\"items_per_thread\": %d, \"cap\": %d, \"xcd_swizzle\": %d, \"min_piece\": %.3f, \"min_entries\": %lld}", first_rule ? "" : ", ",
                             kDtypeNames[di], r.items_per_thread, r.cap, r.xcd_swizzle, r.min_piece, (long long)r.min_entries);
                first_rule = false;
            }
        std::fprintf(f, "},\n");
    }
    std::fprintf(f, "  \"entries\": [\n");
    bool first = true;
    for (int fi = 0; fi < CMI_TABLE_KEYS; fi++)
        for (int di = 0; di < 2; di++)
            for (int b = 0; b < kBuckets; b++) {
                if (!g_table.valid[fi][di][b]) continue;
                const cmi_config &c = g_table.cfg[fi][di][b];
                std::fprintf(f,
                             "%s    {\"format\": \"%s\", \"dtype\": \"%s\", \"bucket\": %d, \"kernel\": %d, "
                             "\"block_size\": %d, \"threads_per_row\": %d, \"rows_per_block\": %d, "
                             "\"items_per_thread\": %d, \"nontemporal\": %d, \"xcd_swizzle\": %d, \"blocks_per_cu\": %d, "
                             "\"mean\": %.4f}",
                             first ? "" : ",\n", kFormatNames[fi], kDtypeNames[di], b, c.kernel, c.block_size,
                             c.threads_per_row, c.rows_per_block, c.items_per_thread, c.nontemporal, c.xcd_swizzle,
                             c.blocks_per_cu, g_table.mean[fi][di][b]);
                first = false;
            }
    std::fprintf(f, "\n  ]\n}\n");
    std::fclose(f);
    return CMI_SUCCESS;
}

CMI_API int cmi_tuning_clear(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(&g_table, 0, sizeof(g_table));
    g_default_loaded = true; // stay on the built-in heuristics until the next explicit load
    return CMI_SUCCESS;
}

CMI_API int cmi_tuning_set(int format, int dtype, double mean_entries_per_row, const cmi_config *cfg)
{
    if (format < 0 || format >= CMI_TABLE_KEYS || dtype < 0 || dtype > 1 || !cfg)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_set: bad format/dtype/config");
    if (format == CMI_FORMAT_COO && cfg->kernel == CMI_COO_TILE) // a plan-less multiply runs this key on entries in ANY order
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_set: CMI_COO_TILE needs row-sorted entries: it belongs under CMI_TABLE_COO_SORTED");
    if (cfg->kernel == CMI_CSR_STREAM_C16) // needs a plan's 16-bit copy: a plan-less multiply could not run it
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_set: CMI_CSR_STREAM_C16 is asked for per plan (cmi_plan_create_csr), not per table entry");
    std::lock_guard<std::mutex> lk(g_mu);
    ensure_default_loaded(); // explicit entries layer ON TOP of the shipped table: a later AUTO multiply must not reload over them
    const int b = bucket_of(mean_entries_per_row);
    g_table.cfg[format][dtype][b] = *cfg;
    g_table.valid[format][dtype][b] = true;
    g_table.mean[format][dtype][b] = mean_entries_per_row > 0.0 ? mean_entries_per_row : 0.0;
    return CMI_SUCCESS;
}

CMI_API int cmi_tuning_hyb_rule(int dtype, int *kind, double *relative_speed, int64_t *threshold)
{
    if (dtype < 0 || dtype > 1) return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_hyb_rule: bad value type");
    std::lock_guard<std::mutex> lk(g_mu);
    ensure_default_loaded();
    const bool v = g_table.hyb_valid[dtype];
    if (kind) *kind = v ? g_table.hyb_kind[dtype] : CMI_HYB_RULE_REFERENCE;
    if (relative_speed) *relative_speed = v ? g_table.hyb_relative_speed[dtype] : kRefRelativeSpeed;
    if (threshold) *threshold = v ? g_table.hyb_threshold[dtype] : kRefBreakeven;
    return CMI_SUCCESS;
}

CMI_API int cmi_tuning_set_hyb_rule(int dtype, int kind, double relative_speed, int64_t threshold)
{
    if (dtype < 0 || dtype > 1 || kind < CMI_HYB_RULE_REFERENCE || kind > CMI_HYB_RULE_COST2 || !(relative_speed > 0.0) || threshold < 0)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_set_hyb_rule: bad value type, kind, speed or threshold");
    std::lock_guard<std::mutex> lk(g_mu);
    ensure_default_loaded();
    if (!g_table.hyb_valid[dtype]) g_table.hyb_light_speed[dtype] = relative_speed;
    g_table.hyb_valid[dtype] = true;
    g_table.hyb_kind[dtype] = kind;
    g_table.hyb_relative_speed[dtype] = relative_speed;
    g_table.hyb_threshold[dtype] = (long)threshold;
    return CMI_SUCCESS;
}

// the fourth parameter of CMI_HYB_RULE_COST2 (ignored by the other kinds): cost of a COO entry, in ELL slots, while the COO part is
// light enough for the one-launch kernel
// csr_waver's launch shape and AUTO gates (plan.hip waver_try): the table's rule, else the built-in defaults
namespace cmi {
void waver_rule(int dtype, cmi_waver_rule *out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    ensure_default_loaded();
    *out = g_table.waver_valid[dtype] ? g_table.waver[dtype] : default_waver_rule(dtype);
}
} // namespace cmi
CMI_API int cmi_tuning_waver_rule(int dtype, cmi_waver_rule *rule)
{
    if (dtype < 0 || dtype > 1 || !rule) return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_waver_rule: bad argument");
    cmi::waver_rule(dtype, rule);
    return CMI_SUCCESS;
}
CMI_API int cmi_tuning_set_waver_rule(int dtype, const cmi_waver_rule *rule)
{
    if (dtype < 0 || dtype > 1 || !rule || (rule->items_per_thread != 1 && rule->items_per_thread != 2 && rule->items_per_thread != 4) ||
        (rule->cap != 0 && rule->cap != 3 && rule->cap != 4) || rule->xcd_swizzle < 0 || !(rule->min_piece >= 1.0) || rule->min_entries < 0)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_set_waver_rule: items_per_thread 1 / 2 / 4, cap 0 / 3 / 4, xcd_swizzle >= 0, min_piece >= 1, min_entries >= 0");
    std::lock_guard<std::mutex> lk(g_mu);
    ensure_default_loaded();
    g_table.waver[dtype] = *rule;
    g_table.waver_valid[dtype] = true;
    return CMI_SUCCESS;
}

CMI_API int cmi_tuning_hyb_light_speed(int dtype, double *light_speed)
{
    if (dtype < 0 || dtype > 1 || !light_speed) return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_hyb_light_speed: bad argument");
    std::lock_guard<std::mutex> lk(g_mu);
    ensure_default_loaded();
    *light_speed = g_table.hyb_valid[dtype] ? g_table.hyb_light_speed[dtype] : kRefRelativeSpeed;
    return CMI_SUCCESS;
}
CMI_API int cmi_tuning_set_hyb_light_speed(int dtype, double light_speed)
{
    if (dtype < 0 || dtype > 1 || !(light_speed > 0.0)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_set_hyb_light_speed: bad argument");
    std::lock_guard<std::mutex> lk(g_mu);
    ensure_default_loaded();
    if (!g_table.hyb_valid[dtype]) { // a light speed alone: the reference's rule stays until a kind is set
        g_table.hyb_valid[dtype] = true;
        g_table.hyb_kind[dtype] = CMI_HYB_RULE_REFERENCE;
        g_table.hyb_relative_speed[dtype] = kRefRelativeSpeed;
        g_table.hyb_threshold[dtype] = kRefBreakeven;
    }
    g_table.hyb_light_speed[dtype] = light_speed;
    return CMI_SUCCESS;
}

CMI_API int cmi_tuning_select(int format, int dtype, int64_t num_rows, int64_t num_cols, int64_t num_entries,
                              cmi_config *out)
{
    if (format < 0 || format >= CMI_TABLE_KEYS || dtype < 0 || dtype > 1 || !out)
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_tuning_select: bad format/dtype/out");
    select_config(format, dtype, num_rows, num_cols, num_entries, nullptr, out);
    return CMI_SUCCESS;
}
