/*
 * synth_gen.c -- C twin of the deterministic integer-only IQ generator of SURVEY.md section 8d
 * (device: rtlamr_amd/csrc/synth.h, numpy: rtlamr_amd/synth.py).
 *
 * TEST INFRASTRUCTURE ONLY, like the rest of oracle/: it lets the CPU oracle produce golden
 * results for the BASELINE.json workloads at their full sizes (tests/golden/make_bench_golden.py)
 * on streams that never saw a GPU.  tests/test_oracle_golden.py compares it byte for byte with
 * the numpy generator; the GPU tests compare the device generator with the same.
 *
 *   h = splitmix64(seed ^ n);  I = 119 + popcount(h & 0xFFFF);  Q = 120 + popcount((h >> 16) & 0xFFFF)
 * Packets are Manchester-OOK bursts: bit 1 = chip high then low, bit 0 = low then high (the sign
 * convention of Decoder.Filter, protocol/decode.go:239-244); "high" adds (dI, dQ), clamped to [0, 255].
 */
#include <stdint.h>
#include <stddef.h>

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* out[2*n_samples] interleaved I,Q of stream samples [first_sample, first_sample + n_samples) */
void orc_synth_noise(uint8_t *out, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    for (uint64_t i = 0; i < n_samples; i++) {
        const uint64_t h = splitmix64(seed ^ (first_sample + i));
        out[2 * i] = (uint8_t)(119 + __builtin_popcount((uint32_t)h & 0xFFFFu));
        out[2 * i + 1] = (uint8_t)(120 + __builtin_popcount((uint32_t)(h >> 16) & 0xFFFFu));
    }
}

/* the second distribution of SURVEY.md 8d: uniform random bytes (device twin: k_synth_noise<true>) */
void orc_synth_uniform(uint8_t *out, uint64_t n_samples, uint64_t seed, uint64_t first_sample)
{
    for (uint64_t i = 0; i < n_samples; i++) {
        const uint64_t h = splitmix64(seed ^ (first_sample + i));
        out[2 * i] = (uint8_t)(h >> 32);
        out[2 * i + 1] = (uint8_t)(h >> 40);
    }
}

/*
 * One packet: starts at stream sample `start`, n_bits bits (MSB first inside each byte of `bits`).
 * iq holds stream samples [first_sample, first_sample + n_samples); samples outside are skipped.
 */
void orc_synth_plant(uint8_t *iq, uint64_t n_samples, uint64_t first_sample, int chip_length,
                     uint64_t start, const uint8_t *bits, uint32_t n_bits, int d_i, int d_q)
{
    const uint32_t sl = 2u * (uint32_t)chip_length;
    for (uint32_t p = 0; p < n_bits; p++) {
        const uint32_t bit = (bits[p >> 3] >> (7 - (p & 7))) & 1u;
        /* the high chip: first half of the symbol for a 1, second half for a 0 */
        const uint64_t s0 = start + (uint64_t)p * sl + (bit ? 0u : (uint32_t)chip_length);
        for (uint32_t k = 0; k < (uint32_t)chip_length; k++) {
            const uint64_t n = s0 + k;
            if (n < first_sample || n >= first_sample + n_samples) continue;
            uint8_t *px = iq + 2 * (n - first_sample);
            const int I = (int)px[0] + d_i, Q = (int)px[1] + d_q;
            px[0] = (uint8_t)(I < 0 ? 0 : I > 255 ? 255 : I);
            px[1] = (uint8_t)(Q < 0 ? 0 : Q > 255 ? 255 : Q);
        }
    }
}
