// tests/cpp/fuzz_ljpeg.cpp -- the lossless-JPEG decoder (raweditor_amd/csrc/rd_ljpeg.h) under AddressSanitizer + UBSan on a
// corpus of damaged streams: records are [u32 length][bytes].  Any out-of-bounds access aborts the process.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Iraweditor_amd/csrc tests/cpp/fuzz_ljpeg.cpp -o /tmp/fuzz_ljpeg
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rd_ljpeg.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    unsigned ok = 0, err = 0;
    for (;;) {
        uint32_t len;
        if (fread(&len, 4, 1, f) != 1) break;
        std::vector<uint8_t> buf(len);
        if (len && fread(buf.data(), 1, len, f) != len) break;
        const size_t cap = 4096;                                  // deliberately small: larger frames must be refused, not written
        std::vector<uint16_t> dst(cap);
        uint32_t w = 0, h = 0, nc = 0, p = 0;
        const int rc = rd_ljpeg::decode(buf.data(), buf.size(), dst.data(), cap, &w, &h, &nc, &p);
        if (rc == rd_ljpeg::OK) ++ok; else ++err;
    }
    fclose(f);
    printf("decoded %u, refused %u\n", ok, err);
    return 0;
}
