// ref_harness.cpp -- C entry points around the MECHANICALLY TRANSLATED reference (oracle/_ref/ref_gen.hpp, generated
// by go2cxx.py from /root/reference's Go sources at build time; never committed).
//
// TEST INFRASTRUCTURE.  This file is a CALLER of the reference's public API, written by hand like
// go/cmd/amdgolden/main.go: NewDecoder, NewParser of each protocol package, RegisterProtocol, Allocate, then Decode
// once per block with the message channel drained until closed (main.go:64-86, 235-292).  What Decode hands to the
// parsers (the []Data of decode.go:178) is observed through the API too: a "spy" protocol.Parser with the same
// configuration as the real one is registered next to it, so it lands in the same preamble group and receives the
// same slice.  Nothing here restates the algorithm; every number it returns was computed by translated reference code.
// Only tests/ load the resulting library (tests/test_ref_translated.py); nothing in rtlamr_amd/ or bench.py does.
#include "ref_gen.hpp"

#include <cstdint>
#include <string>
#include <vector>

namespace {

using go::Slice;
using go::String;

struct Spy {
    P_protocol::PacketConfig cfg;
    std::mutex mu;
    std::vector<long long> idx;
    std::vector<std::vector<uint8_t>> bytes;
    std::vector<std::string> bits;
    void Parse(Slice<P_protocol::Data> pkts, go::Chan<P_protocol::Message>, P_sync::WaitGroup* wg) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (long long i = 0; i < pkts.len; i++) {
                const P_protocol::Data& d = pkts.ptr[i];
                idx.push_back(d.Idx.v);
                bytes.emplace_back((const uint8_t*)d.Bytes.ptr, (const uint8_t*)d.Bytes.ptr + d.Bytes.len);
                bits.push_back(d.Bits.str());
            }
        }
        wg->Done();
    }
    void SetDecoder(P_protocol::Decoder*) {}
    P_protocol::PacketConfig Cfg() { return cfg; }
};

struct Msg {
    std::string type, checksum_hex, record;
    unsigned long long id, mtype;
};

struct Ref {
    P_protocol::Decoder d;
    std::vector<Spy*> spies;  // one per distinct preamble, in registration order
    std::vector<Msg> msgs;
    P_r900::Parser* r900 = nullptr;
    long long calls = 0;
};

P_protocol::Parser make_parser(const std::string& name, int chip, Ref* r) {
    go::int_ cl((long long)chip);
    if (name == "scm") return P_scm::NewParser(cl);
    if (name == "scm+") return P_scmplus::NewParser(cl);
    if (name == "idm") return P_idm::NewParser(cl);
    if (name == "netidm") return P_netidm::NewParser(cl);
    if (name == "r900") {
        P_protocol::Parser p = P_r900::NewParser(cl);
        r->r900 = go::type_assert<P_r900::Parser*>(p);
        return p;
    }
    if (name == "r900bcd") return P_r900bcd::NewParser(cl);   // wraps r900.NewParser (r900bcd.go:35-37); its r900 buffers are not exposed
    return P_protocol::Parser();
}

std::string hex(const Slice<go::byte>& b) {
    static const char* d = "0123456789abcdef";
    std::string o;
    for (long long i = 0; i < b.len; i++) {
        o += d[b.ptr[i].v >> 4];
        o += d[b.ptr[i].v & 15];
    }
    return o;
}

void one_call(Ref* r, const uint8_t* block, long long nbytes) {
    for (Spy* s : r->spies) {
        s->idx.clear();
        s->bytes.clear();
        s->bits.clear();
    }
    r->msgs.clear();
    Slice<go::byte> input((go::byte*)block, nbytes, nbytes);
    auto ch = r->d.Decode(input);
    for (;;) {  // for msg := range d.Decode(block)  (main.go:235)
        auto [m, ok] = ch.recv2();
        if (!ok) break;
        Msg x;
        x.type = m.MsgType().str();
        x.id = m.MeterID().v;
        x.mtype = m.MeterType().v;
        x.checksum_hex = hex(m.Checksum());
        auto rec = m.Record();
        for (long long i = 0; i < rec.len; i++) x.record += (i ? "," : "") + rec.ptr[i].str();
        r->msgs.push_back(x);
    }
    r->calls++;
}

}  // namespace

extern "C" {

void* ref_new(const char* protocols_csv, int chip) {
    Ref* r = new Ref();
    r->d = P_protocol::NewDecoder();
    std::string all(protocols_csv), name;
    std::vector<std::string> names;
    for (char c : all + ",") {
        if (c == ',') {
            if (!name.empty()) names.push_back(name);
            name.clear();
        } else name += c;
    }
    for (const std::string& nm : names) {
        P_protocol::Parser p = make_parser(nm, chip, r);
        if (p == go::nil) {
            delete r;
            return nullptr;
        }
        r->d.RegisterProtocol(p);
        std::string pre = p.Cfg().Preamble.str();
        bool have = false;
        for (Spy* s : r->spies) have = have || s->cfg.Preamble.str() == pre;
        if (!have) {
            Spy* s = new Spy();
            s->cfg = p.Cfg();
            r->spies.push_back(s);
            r->d.RegisterProtocol(P_protocol::Parser(s));
        }
    }
    r->d.Allocate();
    return r;
}

void ref_free(void* h) { delete (Ref*)h; }

// DataRate, ChipLength, SymbolLength, SampleRate, PreambleSymbols, PacketSymbols, PreambleLength, PacketLength,
// BlockSize, BlockSize2, BufferLength, number of distinct preambles, CenterFreq, len(Signal), len(Quantized), pkt bytes
void ref_geometry(void* h, int64_t* out) {
    Ref* r = (Ref*)h;
    const auto& c = r->d.Cfg;
    long long v[16] = {c.DataRate.v, c.ChipLength.v, c.SymbolLength.v, c.SampleRate.v, c.PreambleSymbols.v, c.PacketSymbols.v,
                       c.PreambleLength.v, c.PacketLength.v, c.BlockSize.v, c.BlockSize2.v, c.BufferLength.v,
                       (long long)r->spies.size(), (long long)c.CenterFreq.v, r->d.Signal.len, r->d.Quantized.len, r->d.pkt.len};
    for (int i = 0; i < 16; i++) out[i] = v[i];
}

int64_t ref_decode(void* h, const uint8_t* block, int64_t nbytes) {
    Ref* r = (Ref*)h;
    one_call(r, block, nbytes);
    return (int64_t)r->msgs.size();
}

int64_t ref_copy_quantized(void* h, uint8_t* out) {
    Ref* r = (Ref*)h;
    std::memcpy(out, (const void*)r->d.Quantized.ptr, (size_t)r->d.Quantized.len);
    return r->d.Quantized.len;
}

int64_t ref_copy_signal(void* h, float* out) {
    Ref* r = (Ref*)h;
    std::memcpy(out, (const void*)r->d.Signal.ptr, (size_t)r->d.Signal.len * 4);
    return r->d.Signal.len;
}

int64_t ref_spy_preamble(void* h, int i, char* out, int64_t cap) {
    Ref* r = (Ref*)h;
    std::string s = r->spies[i]->cfg.Preamble.str();
    if ((int64_t)s.size() + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}

// hits the last call handed to the parsers of preamble i: idx, packet bytes (row width = len(d.pkt)); returns count
int64_t ref_spy_hits(void* h, int i, int64_t* idx, uint8_t* bytes, int64_t cap) {
    Ref* r = (Ref*)h;
    Spy* s = r->spies[i];
    long long w = r->d.pkt.len;
    long long n = (long long)s->idx.size();
    for (long long k = 0; k < n && k < cap; k++) {
        idx[k] = s->idx[k];
        if ((long long)s->bytes[k].size() != w) return -2;
        std::memcpy(bytes + k * w, s->bytes[k].data(), (size_t)w);
    }
    return n;
}

// Data.Bits of hit k of preamble i (parse.go:61-69)
int64_t ref_spy_bits(void* h, int i, int64_t k, char* out, int64_t cap) {
    Ref* r = (Ref*)h;
    const std::string& s = r->spies[i]->bits[k];
    if ((int64_t)s.size() + 1 > cap) return -1;
    std::memcpy(out, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}

// n consecutive Decode calls.  q: packed (MSB first) Quantized[PacketLength:] of every call; hits rows (block, preamble
// index, idx) in (block, preamble, idx) order with their packet bytes.  Returns the number of hits (may exceed cap:
// rows beyond cap are dropped).  msgs (optional): one text line per message, "call|MsgType|MeterID|MeterType|checksum|record"
int64_t ref_decode_stream(void* h, const uint8_t* iq, int64_t n_blocks, uint8_t* q, int32_t* hits, uint8_t* hit_bytes,
                          int64_t cap, char* msgs, int64_t msgs_cap, int64_t* msgs_len) {
    Ref* r = (Ref*)h;
    const long long bs = r->d.Cfg.BlockSize.v, bs2 = r->d.Cfg.BlockSize2.v, pl = r->d.Cfg.PacketLength.v, w = r->d.pkt.len;
    long long total = 0;
    std::string text;
    for (long long k = 0; k < n_blocks; k++) {
        one_call(r, iq + k * bs2, bs2);
        if (q) {
            const go::byte* src = r->d.Quantized.ptr + pl;
            for (long long b = 0; b < bs / 8; b++) {
                unsigned v = 0;
                for (int j = 0; j < 8; j++) v = (v << 1) | src[b * 8 + j].v;
                q[k * (bs / 8) + b] = (uint8_t)v;
            }
        }
        for (size_t p = 0; p < r->spies.size(); p++) {
            Spy* s = r->spies[p];
            for (size_t t = 0; t < s->idx.size(); t++, total++) {
                if (total >= cap) continue;
                hits[total * 3 + 0] = (int32_t)k;
                hits[total * 3 + 1] = (int32_t)p;
                hits[total * 3 + 2] = (int32_t)s->idx[t];
                std::memcpy(hit_bytes + total * w, s->bytes[t].data(), (size_t)w);
            }
        }
        for (const Msg& m : r->msgs)
            text += std::to_string(k) + "|" + m.type + "|" + std::to_string(m.id) + "|" + std::to_string(m.mtype) + "|" + m.checksum_hex + "|" + m.record + "\n";
    }
    if (msgs_len) *msgs_len = (int64_t)text.size();
    if (msgs && (int64_t)text.size() + 1 <= msgs_cap) std::memcpy(msgs, text.c_str(), text.size() + 1);
    return total;
}

// messages of the last ref_decode call, same text form
int64_t ref_messages(void* h, char* out, int64_t cap) {
    Ref* r = (Ref*)h;
    std::string text;
    for (const Msg& m : r->msgs)
        text += std::to_string(r->calls - 1) + "|" + m.type + "|" + std::to_string(m.id) + "|" + std::to_string(m.mtype) + "|" + m.checksum_hex + "|" + m.record + "\n";
    if ((int64_t)text.size() + 1 > cap) return -1;
    std::memcpy(out, text.c_str(), text.size() + 1);
    return (int64_t)text.size();
}

// the r900 parser's own buffers after the last call (r900.go:48-50; filled by filter(), r900.go:82-150)
int64_t ref_r900_quantized(void* h, uint8_t* out, int64_t cap) {
    Ref* r = (Ref*)h;
    if (!r->r900 || r->r900->quantized.len > cap) return -1;
    std::memcpy(out, (const void*)r->r900->quantized.ptr, (size_t)r->r900->quantized.len);
    return r->r900->quantized.len;
}
int64_t ref_r900_signal(void* h, float* out, int64_t cap) {
    Ref* r = (Ref*)h;
    if (!r->r900 || r->r900->signal.len > cap) return -1;
    std::memcpy(out, (const void*)r->r900->signal.ptr, (size_t)r->r900->signal.len * 4);
    return r->r900->signal.len;
}

void ref_lut(float* out256) {
    P_protocol::MagLUT lut = P_protocol::NewMagLUT();
    for (int i = 0; i < 256; i++) out256[i] = lut.ptr[i].v;
}

int64_t ref_next_power_of_2(int64_t v) { return P_protocol::NextPowerOf2(go::int_((long long)v)).v; }

// crc.Checksum through the translated table code (crc/crc.go:34-55)
uint32_t ref_crc(uint32_t init, uint32_t poly, const uint8_t* data, int64_t n) {
    P_crc::CRC c = P_crc::NewCRC(String("x"), go::uint16((uint16_t)init), go::uint16((uint16_t)poly), go::uint16((uint16_t)0));
    return c.Checksum(Slice<go::byte>((go::byte*)data, n, n)).v;
}

}  // extern "C"
