#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/odise_hip.h"
// deterministic mutation fuzzing of the host half of the JPEG decoder under AddressSanitizer
static uint64_t rng = 88172645463325252ull;
static inline uint64_t next() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
int main(int argc, char** argv) {
    const int iters = getenv("FUZZ_ITERS") ? atoi(getenv("FUZZ_ITERS")) : 4000;
    long total = 0, ok = 0;
    for (int f = 1; f < argc; ++f) {
        FILE* fp = fopen(argv[f], "rb");
        if (!fp) continue;
        std::vector<uint8_t> base;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) base.insert(base.end(), buf, buf + n);
        fclose(fp);
        for (int it = 0; it < iters; ++it) {
            std::vector<uint8_t> d = base;
            const int muts = 1 + next() % 8;
            for (int m = 0; m < muts; ++m) {
                const uint64_t r = next();
                const size_t pos = r % d.size();
                switch ((r >> 32) % 5) {
                    case 0: d[pos] = (uint8_t)(r >> 40); break;
                    case 1: d[pos] = 0xFF; break;
                    case 2: d.resize(pos + 1); break;                       // truncate
                    case 3: d.insert(d.begin() + pos, (uint8_t)(r >> 48)); break;
                    default: if (d.size() > 8) d.erase(d.begin() + pos); break;
                }
                if (d.empty()) d.push_back(0);
            }
            // exact-size heap copy so that any read past the end is caught
            uint8_t* exact = (uint8_t*)malloc(d.size());
            memcpy(exact, d.data(), d.size());
            odise_jpeg_info info;
            ++total;
            if (odise_hip_jpeg_info(exact, (int64_t)d.size(), &info) == 0 && info.coef_count < (1 << 24)) {
                std::vector<int16_t> coefs((size_t)info.coef_count);
                uint16_t qt[3 * 64];
                if (odise_hip_jpeg_entropy_decode(exact, (int64_t)d.size(), coefs.data(), info.coef_count, qt) == 0) ++ok;
            }
            free(exact);
        }
    }
    printf("%ld mutated streams, %ld decoded without an error code\n", total, ok);
    return 0;
}
