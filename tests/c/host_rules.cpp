// Host-side arithmetic of the kernel headers that decides launch shapes (no device needed): compiled by
// tests/test_host_rules.py with hipcc, prints one line per case for the test to compare.
#include <cstdio>
#include "k3_slice.h"

int main()
{
    using namespace amr;
    // k3_fold(n_tiles, n_pre, slots): fold the history tile into workgroup 0 only where that saves a round of the chip
    struct { uint32_t t, p, s; } f[] = {{2049, 1, 2048}, {2048, 1, 2048}, {2, 1, 2048}, {2, 4, 2048}, {8193, 4, 2048}, {4097, 1, 2048},
                                        {1025, 2, 2048}, {1, 1, 2048}, {2049, 1, 1024}, {1564, 1, 2048}};
    for (auto &c : f) printf("fold %u %u %u = %d\n", c.t, c.p, c.s, (int)k3_fold(c.t, c.p, c.s));
    // group sums: one cache line each
    printf("stride %u groups(2049) %u groups(64) %u groups(65) %u\n", kGroupStride, k2_groups(2049), k2_groups(64), k2_groups(65));
    // dynamic LDS of K3: rows of a long packet, or 257 short packets for K5's test
    SearchGeom g{};
    g.block_size = 4096; g.lg_block_size = 12; g.wpb = 128; g.lg_wpb = 7; g.symbol_length = 144; g.packet_symbols = 96; g.pkt_bytes = 12;
    printf("lds scm72 plain %zu validated %zu\n", k3_lds_bytes(g, false), k3_lds_bytes(g, true));
    g.block_size = 8192; g.lg_block_size = 13; g.wpb = 256; g.lg_wpb = 8; g.packet_symbols = 736; g.pkt_bytes = 92;
    printf("lds idm72 plain %zu validated %zu\n", k3_lds_bytes(g, false), k3_lds_bytes(g, true));
    return 0;
}
