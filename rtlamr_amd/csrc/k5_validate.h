// K5 -- per-hit validation on the GPU: the checksum test every rtlamr parser applies first and
// the removal of repeated packets, so that only hits a parser could turn into a message leave the device.
//
// Reference semantics restated (the parsers still run unchanged on what is left and emit the same messages):
//   crc.Checksum            crc/crc.go:49-55   crc = crc<<8 ^ table[crc>>8 ^ byte], table from crc/crc.go:34-47
//   SCM    Parse            scm/scm.go:61-90   Checksum(Bytes[2:12]) != 0        -> skip     (BCH 0x6F63, init 0)
//   SCM+   Parse            scmplus/scmplus.go:62-92   Checksum(Bytes[2:16]) != Residue -> skip  (CCITT)
//   IDM / NetIDM Parse      idm/idm.go:62-98, netidm/netidm.go:73-110
//                           Checksum(Bytes[4:92]) != Residue -> skip; Checksum(Bytes[9:13] ++ Bytes[88:90]) != Residue -> skip
//   seen[string(Bytes)]     first line of every Parse loop: a byte string is handled once per Decode call.
// A hit is dropped here when a check fails, or when its first dedupe_bytes packet bytes equal those of the hit
// right before it in the same (preamble, block) list -- a subset of what `seen` drops, so the parser's own `seen`
// finishes the job and the message stream is unchanged.  The order of the surviving hits is kept.
//
// Where it runs (round 4): the per-hit test is the LAST STAGE OF K3's workgroup -- a workgroup owns one (tile, preamble)
// list, has just written its packets and tests them while the following batch's demodulation still holds the chip --
// and one kernel of the same grid (k5_compact, k3_slice.h) moves the survivors into a second packed buffer of the same
// layout as K3's.  Rounds 2-3 ran two kernels over chunks of 256 hits behind K3: with the stream's ~5 us between
// dependent launches the validated tail of a batch (45 us) outlasted the search of the next one (29 us) and every
// step of a multi-GPU run -- validated hits are what the ranks gather -- paid the difference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "amrdemod.h"

namespace amr {

struct ValCheck {
    uint16_t init, poly, residue, n_spans;
    uint16_t off[2], len[2];
};
struct ValRule {
    int32_t n_checks;       // 0: no checksum test
    int32_t dedupe_bytes;   // 0: keep repeated packets
    ValCheck chk[2];
};

// crc.NewTable (crc/crc.go:34-47) for the checks of one rule, by 256 threads: tbl[c][byte]
__device__ __forceinline__ void k5_tables(const ValRule &r, uint16_t (*tbl)[256], uint32_t tid)
{
    for (int c = 0; c < r.n_checks; ++c)
        for (uint32_t t = tid; t < 256; t += 256) {
            const uint16_t poly = r.chk[c].poly;
            uint16_t crc = (uint16_t)(t << 8);
            for (int b = 0; b < 8; ++b) crc = (crc & 0x8000) ? (uint16_t)((crc << 1) ^ poly) : (uint16_t)(crc << 1);
            tbl[c][t] = crc;
        }
}

// every check of the rule on one packet (crc/crc.go:49-55)
__device__ __forceinline__ bool k5_checks(const ValRule &r, const uint16_t (*tbl)[256], const uint8_t *pkt)
{
    bool keep = true;
    for (int c = 0; c < r.n_checks; ++c) {
        const ValCheck &k = r.chk[c];
        uint16_t crc = k.init;
        for (uint32_t s = 0; s < k.n_spans; ++s)
#pragma clang loop unroll_count(4)
            for (uint32_t i = 0; i < k.len[s]; ++i)
                crc = (uint16_t)((crc << 8) ^ tbl[c][(crc >> 8) ^ pkt[k.off[s] + i]]);   // crc/crc.go:52
        keep = keep && crc == k.residue;
    }
    return keep;
}

}  // namespace amr
