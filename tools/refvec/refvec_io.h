// refvec_io.h — writer of the reference-vector container ("ORBVEC01") that tests/test_reference_vectors.py consumes.
// Plain C++98-compatible, no OpenCV: shared by tools/refvec/dump_reference_vectors.cc (compiled by a maintainer against
// OpenCV + the reference's own sources) and tools/refvec/refvec_selftest.cc (compiled here to pin the format against
// the Python reader oracle/refvec.py).
//
// File = 8-byte magic "ORBVEC01", then records until EOF, all little-endian:
//     u32 name_len | name bytes | u32 dtype (0 u8, 1 i32, 2 f32, 3 f64) | u32 ndim | u64 dims[ndim] | raw data (C order)
// Names are "case/key" or "case/L<level>/key" (see tests/golden/README.md for the key list).
#ifndef ORBX_REFVEC_IO_H
#define ORBX_REFVEC_IO_H
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

namespace refvec {

enum DType { U8 = 0, I32 = 1, F32 = 2, F64 = 3 };
static inline size_t dtype_size(int dt) { return dt == U8 ? 1 : dt == F64 ? 8 : 4; }

class Writer {
public:
    explicit Writer(const char *path) : f_(fopen(path, "wb")) {
        if (f_) fwrite("ORBVEC01", 1, 8, f_);
    }
    ~Writer() { if (f_) fclose(f_); }
    bool ok() const { return f_ != NULL; }
    // data: prod(dims) elements of dtype, C order
    void put(const std::string &name, int dtype, const std::vector<uint64_t> &dims, const void *data) {
        if (!f_) return;
        const uint32_t nl = (uint32_t)name.size(), dt = (uint32_t)dtype, nd = (uint32_t)dims.size();
        uint64_t n = 1;
        for (size_t i = 0; i < dims.size(); i++) n *= dims[i];
        fwrite(&nl, 4, 1, f_);
        fwrite(name.data(), 1, nl, f_);
        fwrite(&dt, 4, 1, f_);
        fwrite(&nd, 4, 1, f_);
        if (nd) fwrite(&dims[0], 8, nd, f_);
        if (n) fwrite(data, dtype_size(dtype), (size_t)n, f_);
    }
    void put1d(const std::string &name, int dtype, uint64_t n, const void *data) {
        std::vector<uint64_t> d(1, n);
        put(name, dtype, d, data);
    }
    void put2d(const std::string &name, int dtype, uint64_t rows, uint64_t cols, const void *data) {
        std::vector<uint64_t> d(2);
        d[0] = rows; d[1] = cols;
        put(name, dtype, d, data);
    }
    void put_text(const std::string &name, const std::string &s) { put1d(name, U8, s.size(), s.data()); }
private:
    FILE *f_;
    Writer(const Writer &);
    Writer &operator=(const Writer &);
};

// CRC-32 (IEEE 802.3, the polynomial of zlib.crc32) of a strided 8-bit image: how the lean form pins pixel arrays
static inline uint32_t crc32_rows(const uint8_t *p, int rows, int cols, size_t stride, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (int r = 0; r < rows; r++) {
        const uint8_t *q = p + (size_t)r * stride;
        for (int c = 0; c < cols; c++) crc = table[(crc ^ q[c]) & 0xFF] ^ (crc >> 8);
    }
    return ~crc;
}

// Minimal binary PGM (P5, maxval 255) reader: the inputs come from tools/refvec/write_refvec_inputs.py, byte for byte
static inline bool read_pgm(const char *path, std::vector<uint8_t> &pix, int &w, int &h) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int maxv = 0;
    bool ok = fscanf(f, "%2s", magic) == 1 && strcmp(magic, "P5") == 0 && fscanf(f, "%d %d %d", &w, &h, &maxv) == 3 && maxv == 255 &&
              w > 0 && h > 0;
    if (ok) {
        fgetc(f);   // the single whitespace byte behind maxval
        pix.resize((size_t)w * h);
        ok = fread(&pix[0], 1, pix.size(), f) == pix.size();
    }
    fclose(f);
    return ok;
}

}  // namespace refvec
#endif
