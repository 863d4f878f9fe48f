// refvec_selftest.cc — writes a small ORBVEC01 file with tools/refvec/refvec_io.h (the writer the maintainer's dump program
// uses) so that tests/test_reference_vectors.py can pin the format against the Python reader.  No OpenCV, no reference.
//   g++ -std=c++11 -I tools/refvec tools/refvec/refvec_selftest.cc -o /tmp/refvec_selftest && /tmp/refvec_selftest OUT [IN.pgm]
#include "refvec_io.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    refvec::Writer w(argv[1]);
    if (!w.ok()) return 1;
    const uint8_t img[6] = {1, 2, 3, 250, 251, 252};
    const int32_t xyr[6] = {7, 9, 42, -1, 1 << 20, 255};
    const float f[3] = {1.2f, -0.0f, 3.5e-7f};
    const double d[2] = {4294967295.0, 0.5};
    w.put2d("self/image", refvec::U8, 2, 3, img);
    w.put2d("self/L3/fast7", refvec::I32, 2, 3, xyr);
    w.put1d("self/meta_f", refvec::F32, 3, f);
    w.put1d("self/crc", refvec::F64, 2, d);
    w.put2d("self/empty", refvec::I32, 0, 3, NULL);
    w.put_text("self/info", "producer=selftest");
    const double c = (double)refvec::crc32_rows(img, 2, 3, 3);
    w.put1d("self/image_crc", refvec::F64, 1, &c);
    if (argc > 2) {   // PGM reader round trip
        std::vector<uint8_t> pix;
        int pw = 0, ph = 0;
        if (!refvec::read_pgm(argv[2], pix, pw, ph)) return 3;
        w.put2d("self/pgm", refvec::U8, ph, pw, &pix[0]);
        const double pc = (double)refvec::crc32_rows(&pix[0], ph, pw, pw);
        w.put1d("self/pgm_crc", refvec::F64, 1, &pc);
    }
    return 0;
}
