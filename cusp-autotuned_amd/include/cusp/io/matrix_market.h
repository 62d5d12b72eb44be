// cusp/io/matrix_market.h -- read_matrix_market_file / _stream, write_matrix_market_file
// (reference cusp/io/matrix_market.h, cusp/io/detail/matrix_market.inl:160-300):
// "coordinate" storage with real / integer / pattern values (pattern -> 1), general or symmetric
// (off-diagonals mirrored, in file order), 1-based indices validated then made 0-based, the result
// sorted by (row, column), then converted to the requested container.  Unlocks the SuiteSparse
// config of BASELINE.json (nlpkkt120, ldoor, thermal2 are not vendored; read them when supplied).
#pragma once
#include <charconv>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>

#include "../coo_matrix.h"

namespace cusp {
namespace io {

namespace detail {
struct matrix_market_banner { std::string storage, symmetry, type; };

inline void tokenize(std::vector<std::string> &tokens, const std::string &str)
{
    std::istringstream is(str);
    std::string t;
    while (is >> t) tokens.push_back(t);
}

template <typename Stream> void read_banner(Stream &input, matrix_market_banner &banner)
{
    std::string line;
    std::vector<std::string> tokens;
    if (!std::getline(input, line)) throw cusp::io_exception("invalid MatrixMarket banner");
    tokenize(tokens, line);
    if (tokens.size() != 5 || tokens[0] != "%%MatrixMarket" || tokens[1] != "matrix") throw cusp::io_exception("invalid MatrixMarket banner");
    banner.storage = tokens[2];
    banner.type = tokens[3];
    banner.symmetry = tokens[4];
    if (banner.storage != "array" && banner.storage != "coordinate") throw cusp::io_exception("invalid MatrixMarket storage format [" + banner.storage + "]");
    if (banner.type != "complex" && banner.type != "real" && banner.type != "integer" && banner.type != "pattern")
        throw cusp::io_exception("invalid MatrixMarket data type [" + banner.type + "]");
    if (banner.symmetry != "general" && banner.symmetry != "symmetric" && banner.symmetry != "hermitian" && banner.symmetry != "skew-symmetric")
        throw cusp::io_exception("invalid MatrixMarket symmetry [" + banner.symmetry + "]");
}

// "array" storage: `rows cols`, then rows*cols values in column-major order; only real / integer general is
// supported, as in the reference; the dense matrix is then converted to the requested type (zeros dropped)
template <typename Matrix, typename Stream> void read_array_stream(Matrix &mtx, Stream &input, const matrix_market_banner &banner)
{
    typedef typename Matrix::value_type V;
    if (banner.type == "pattern") throw cusp::not_implemented_exception("pattern array MatrixMarket format is not supported");
    if (banner.symmetry != "general") throw cusp::not_implemented_exception("only general array symmetric MatrixMarket format is supported");
    std::string line;
    do {
        if (!std::getline(input, line)) throw cusp::io_exception("unexpected EOF while reading MatrixMarket header");
    } while (line.empty() || line[0] == '%');
    std::vector<std::string> tokens;
    tokenize(tokens, line);
    if (tokens.size() != 2) throw cusp::io_exception("invalid MatrixMarket array format");
    size_t num_rows, num_cols;
    std::istringstream(tokens[0]) >> num_rows;
    std::istringstream(tokens[1]) >> num_cols;
    array2d<V, host_memory, column_major> dense(num_rows, num_cols);
    size_t read = 0;
    double v;
    while (read < num_rows * num_cols && (input >> v)) dense.values[read++] = static_cast<V>(v);
    if (read != num_rows * num_cols) throw cusp::io_exception("unexpected EOF while reading MatrixMarket entries");
    cusp::convert(dense, mtx);
}
} // namespace detail

template <typename Matrix, typename Stream> void read_matrix_market_stream(Matrix &mtx, Stream &input)
{
    typedef typename Matrix::index_type I;
    typedef typename Matrix::value_type V;
    detail::matrix_market_banner banner;
    detail::read_banner(input, banner);
    if (banner.type == "complex") throw cusp::not_implemented_exception("complex MatrixMarket data (real value types only)");
    if (banner.storage == "array") { // dense, column-major (reference matrix_market.inl:344-420, 459-466)
        detail::read_array_stream(mtx, input, banner);
        return;
    }

    std::string line;
    do { // skip comments
        if (!std::getline(input, line)) throw cusp::io_exception("unexpected EOF while reading MatrixMarket header");
    } while (line.empty() || line[0] == '%');
    std::vector<std::string> tokens;
    detail::tokenize(tokens, line);
    if (tokens.size() != 3) throw cusp::io_exception("invalid MatrixMarket coordinate format");
    size_t num_rows, num_cols, num_entries;
    std::istringstream(tokens[0]) >> num_rows;
    std::istringstream(tokens[1]) >> num_cols;
    std::istringstream(tokens[2]) >> num_entries;

    coo_matrix<I, V, host_memory> coo(num_rows, num_cols, num_entries);
    size_t read = 0;
    const bool pattern = banner.type == "pattern";
    long long r, c;
    double v = 1.0;
    // the entries: the rest of the stream in one buffer, parsed with strtoll / strtod -- formatted extraction (`input >> r >> c >> v`, what
    // the reference does) reads 1.7 M entries per second, this 10+ M: nlpkkt120's 50 M lines in seconds instead of half a minute
    std::string rest;
    {
        std::streambuf *sb = input.rdbuf();
        std::vector<char> chunk(1 << 20);
        for (std::streamsize got; (got = sb->sgetn(chunk.data(), (std::streamsize)chunk.size())) > 0;) rest.append(chunk.data(), (size_t)got);
    }
    const char *p = rest.c_str(), *const end = p + rest.size(); // (NUL-terminated: strtod stops there)
    auto skip_space = [&] { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v')) p++; };
    auto integer = [&](long long &out) { // [+-]digits
        skip_space();
        const char *b = p;
        bool neg = false;
        if (p < end && (*p == '-' || *p == '+')) neg = *p++ == '-';
        long long val = 0;
        const char *d = p;
        while (p < end && *p >= '0' && *p <= '9') val = val * 10 + (*p++ - '0');
        if (p == d) { p = b; return false; }
        out = neg ? -val : val;
        return true;
    };
    auto real = [&](double &out) {
        skip_space();
#if defined(__cpp_lib_to_chars) && __cpp_lib_to_chars >= 201611L
        const char *b = p < end && *p == '+' ? p + 1 : p;
        const std::from_chars_result res = std::from_chars(b, end, out);
        if (res.ec == std::errc()) { p = res.ptr; return true; }
        if (res.ec == std::errc::result_out_of_range) { char *q = nullptr; out = std::strtod(p, &q); p = q; return true; } // (inf / denormal: as strtod rounds it)
        return false;
#else
        char *q = nullptr;
        out = std::strtod(p, &q);
        if (q == p) return false;
        p = q;
        return true;
#endif
    };
    while (read < num_entries) {
        if (!integer(r) || !integer(c)) break;
        if (!pattern && !real(v)) break;
        if (r < 1) throw cusp::io_exception("found invalid row index (index < 1)");
        if (c < 1) throw cusp::io_exception("found invalid column index (index < 1)");
        if (static_cast<size_t>(r) > num_rows) throw cusp::io_exception("found invalid row index (index > num_rows)");
        if (static_cast<size_t>(c) > num_cols) throw cusp::io_exception("found invalid column index (index > num_columns)");
        coo.row_indices[read] = static_cast<I>(r - 1);
        coo.column_indices[read] = static_cast<I>(c - 1);
        coo.values[read] = static_cast<V>(v);
        read++;
    }
    if (read != num_entries) throw cusp::io_exception("unexpected EOF while reading MatrixMarket entries");

    if (banner.symmetry != "general") {
        if (banner.symmetry != "symmetric") throw cusp::not_implemented_exception("MatrixMarket I/O does not currently support " + banner.symmetry + " matrices");
        size_t off = 0;
        for (size_t n = 0; n < num_entries; n++) off += coo.row_indices[n] != coo.column_indices[n];
        coo_matrix<I, V, host_memory> general(num_rows, num_cols, num_entries + off);
        size_t nnz = 0;
        for (size_t n = 0; n < num_entries; n++) {
            general.row_indices[nnz] = coo.row_indices[n]; general.column_indices[nnz] = coo.column_indices[n]; general.values[nnz] = coo.values[n]; nnz++;
            if (coo.row_indices[n] != coo.column_indices[n]) {
                general.row_indices[nnz] = coo.column_indices[n]; general.column_indices[nnz] = coo.row_indices[n]; general.values[nnz] = coo.values[n]; nnz++;
            }
        }
        coo.swap(general);
    }
    coo.sort_by_row_and_column();
    cusp::convert(coo, mtx);
}

template <typename Matrix> void read_matrix_market_file(Matrix &mtx, const std::string &filename)
{
    std::ifstream file(filename.c_str());
    if (!file) throw cusp::io_exception(std::string("unable to open file \"") + filename + std::string("\" for reading"));
    read_matrix_market_stream(mtx, file);
}

template <typename Matrix, typename Stream> void write_matrix_market_stream(const Matrix &mtx, Stream &output)
{
    typedef typename Matrix::index_type I;
    typedef typename Matrix::value_type V;
    coo_matrix<I, V, host_memory> coo(mtx);
    output << "%%MatrixMarket matrix coordinate real general\n";
    output << "\t" << coo.num_rows << "\t" << coo.num_cols << "\t" << coo.num_entries << "\n";
    output.precision(17);
    for (size_t n = 0; n < coo.num_entries; n++) output << (coo.row_indices[n] + 1) << " " << (coo.column_indices[n] + 1) << " " << coo.values[n] << "\n";
}

template <typename Matrix> void write_matrix_market_file(const Matrix &mtx, const std::string &filename)
{
    std::ofstream file(filename.c_str());
    if (!file) throw cusp::io_exception(std::string("unable to open file \"") + filename + std::string("\" for writing"));
    write_matrix_market_stream(mtx, file);
}

} // namespace io
} // namespace cusp
