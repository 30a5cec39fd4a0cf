// tools/keccak/bench.cpp -- the library's host Keccak (zk_amd/csrc/keccak_host.hpp: baseline or BMI build, chosen at run time) against the AVX-512
// plane-per-register experiment in k512_experiment.hpp on a 32 MiB stream (the GKR output-layer absorb).  CPU only.
// Build: clang++ -O3 -std=c++17 -Izk_amd/csrc -Itools/keccak tools/keccak/bench.cpp -o /tmp/keccak_bench
#include "keccak_host.hpp"
#include "k512_experiment.hpp"
#include <chrono>
#include <cstdio>
#include <vector>
int main() {
    std::vector<uint8_t> buf((32u << 20) / 136 * 136);
    for (size_t i = 0; i < buf.size(); ++i) buf[i] = (uint8_t)(i * 2654435761u >> 13);
    for (int rep = 0; rep < 3; ++rep) {
        zkh::Keccak256 k;
        auto t0 = std::chrono::steady_clock::now();
        k.update(buf.data(), buf.size());
        uint8_t d[32];
        k.peek(d);
        auto t1 = std::chrono::steady_clock::now();
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        printf("scalar  %.2f ms  %.0f MB/s  digest %02x%02x%02x%02x\n", ms, buf.size() / ms / 1e3, d[0], d[1], d[2], d[3]);
        uint64_t st[25] = {0};
        t0 = std::chrono::steady_clock::now();
        k512::absorb_blocks(st, buf.data(), buf.size() / 136);
        uint8_t last[136] = {0};
        last[0] ^= 0x01; last[135] ^= 0x80;
        k512::absorb_blocks(st, last, 1);
        t1 = std::chrono::steady_clock::now();
        ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        const uint8_t* e = (const uint8_t*)st;
        printf("avx512  %.2f ms  %.0f MB/s  digest %02x%02x%02x%02x\n", ms, buf.size() / ms / 1e3, e[0], e[1], e[2], e[3]);
    }
}
