// Command amdgolden writes golden vectors computed by the REFERENCE decoder
// (github.com/bemasher/rtlamr/protocol, unmodified) in the schema of this
// repository's tests/golden/*.json, so that the CPU oracle and the MI355X
// path can be compared with the Go implementation itself instead of with a
// restatement of it (SURVEY.md section 8, rows c and f4).
//
// It must be built inside a checkout of the reference module, because it
// imports the reference packages:
//
//	cp -r <this repo>/go/cmd/amdgolden  <rtlamr checkout>/cmd/amdgolden
//	cd <rtlamr checkout> && go run ./cmd/amdgolden -repo <this repo>
//
// Output: <repo>/tests/golden/go_sample_bin.json and go_synth.json, and -- when
// built with -tags amdgolden next to go/r900/amd_golden.go (see r900_golden.go)
// -- go_r900_filter.json.  The
// pytest suite picks them up when they exist (tests/test_go_golden.py):
// oracle == Go on every hash, and, on the GPU box, HIP == Go.
//
// What it runs per block is protocol.Decoder.Decode (decode.go:163-183) spelled
// out with the decoder's exported fields and methods -- the history slide,
// MagLUT.Execute, Filter, then Search and Slice per distinct preamble -- so that
// the quantized bits, hit indices and packet bytes can be recorded, which Decode
// itself hands only to the parsers.
//
// The synthetic streams are regenerated here with an integer-only port of
// rtlamr_amd/synth.py and tests/util.py (splitmix64 noise, packet schedule,
// Manchester-OOK planting, CRC-valid packets); every stream's sha256 is checked
// against the iq_sha recorded in tests/golden/synth.json before it is decoded.
package main

import (
	"crypto/sha256"
	"encoding/binary"
	"encoding/hex"
	"encoding/json"
	"flag"
	"fmt"
	"log"
	"math"
	"math/bits"
	"os"
	"path/filepath"
	"sort"

	"github.com/bemasher/rtlamr/crc"
	"github.com/bemasher/rtlamr/protocol"

	_ "github.com/bemasher/rtlamr/idm"
	_ "github.com/bemasher/rtlamr/netidm"
	_ "github.com/bemasher/rtlamr/r900"
	_ "github.com/bemasher/rtlamr/scm"
	_ "github.com/bemasher/rtlamr/scmplus"
)

// ---------------------------------------------------------------- decoder

// refDecoder drives an unmodified protocol.Decoder block by block.
type refDecoder struct {
	d   protocol.Decoder
	lut protocol.MagLUT
	pre [][]byte // distinct preambles as 0/1 bytes, in order of first registration = preamble id

	parsers map[string]protocol.Parser // by protocol name (r900.go: the second-stage golden needs the parser itself)
}

func newRefDecoder(protos []string, chip int) *refDecoder {
	r := &refDecoder{d: protocol.NewDecoder(), lut: protocol.NewMagLUT(), parsers: map[string]protocol.Parser{}}
	seen := map[string]bool{}
	for _, name := range protos {
		p, err := protocol.NewParser(name, chip) // main.go:77
		if err != nil {
			log.Fatal(err)
		}
		r.d.RegisterProtocol(p)
		r.parsers[name] = p
		s := p.Cfg().Preamble
		if !seen[s] {
			seen[s] = true
			b := make([]byte, len(s))
			for i, c := range s {
				if c == '1' {
					b[i] = 1
				}
			}
			r.pre = append(r.pre, b)
		}
	}
	r.d.Allocate()
	return r
}

type hit struct {
	pid, call, idx int
	bytes          []byte
}

// step is one Decode call (decode.go:163-183) without the parsers.
func (r *refDecoder) step(call int, block []byte, q *[]byte, hits *[]hit) {
	cfg := r.d.Cfg
	copy(r.d.Signal, r.d.Signal[cfg.BlockSize:])
	copy(r.d.Quantized, r.d.Quantized[cfg.BlockSize:])
	r.lut.Execute(block, r.d.Signal[cfg.SymbolLength:])
	r.d.Filter(r.d.Signal, r.d.Quantized[cfg.PacketLength:])

	// the BlockSize new decisions, packed MSB first
	fresh := r.d.Quantized[cfg.PacketLength:]
	for i := 0; i < cfg.BlockSize; i += 8 {
		var b byte
		for k := 0; k < 8; k++ {
			b = b<<1 | fresh[i+k]
		}
		*q = append(*q, b)
	}
	for pid, pre := range r.pre {
		for _, data := range r.d.Slice(r.d.Search(pre)) {
			*hits = append(*hits, hit{pid, call, data.Idx, data.Bytes})
		}
	}
}

type result struct {
	q    []byte
	hits []hit
	cfg  protocol.PacketConfig
}

func decodeStream(protos []string, chip int, iq []byte) result {
	r := newRefDecoder(protos, chip)
	bs2 := r.d.Cfg.BlockSize2
	var res result
	res.cfg = r.d.Cfg
	for k := 0; (k+1)*bs2 <= len(iq); k++ {
		r.step(k, iq[k*bs2:(k+1)*bs2], &res.q, &res.hits)
	}
	return res
}

// ---------------------------------------------------------------- hashes

func shaHex(b []byte) string {
	s := sha256.Sum256(b)
	return hex.EncodeToString(s[:])
}

func ones(q []byte) (n int) {
	for _, b := range q {
		n += bits.OnesCount8(b)
	}
	return
}

// hitsSha: rows (preamble id, call, idx) as little-endian int64, sorted by (pid, call, idx)  (tests/util.py oracle_run)
func sortedHits(h []hit) []hit {
	s := append([]hit(nil), h...)
	sort.SliceStable(s, func(a, b int) bool {
		if s[a].pid != s[b].pid {
			return s[a].pid < s[b].pid
		}
		if s[a].call != s[b].call {
			return s[a].call < s[b].call
		}
		return s[a].idx < s[b].idx
	})
	return s
}

func hitsSha(h []hit) string {
	buf := make([]byte, 0, len(h)*24)
	for _, x := range sortedHits(h) {
		for _, v := range []int{x.pid, x.call, x.idx} {
			buf = binary.LittleEndian.AppendUint64(buf, uint64(int64(v)))
		}
	}
	return shaHex(buf)
}

// pktSha: packet bytes of the sorted hits; nbytes = PacketSymbols/8 (whole bytes only) or the full packet
func pktSha(h []hit, sorted bool, nbytes int) string {
	if sorted {
		h = sortedHits(h)
	}
	var buf []byte
	for _, x := range h {
		buf = append(buf, x.bytes[:nbytes]...)
	}
	return shaHex(buf)
}

// ---------------------------------------------------------------- synthetic streams (port of rtlamr_amd/synth.py)

func splitmix64(x uint64) uint64 {
	x += 0x9E3779B97F4A7C15
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9
	x = (x ^ (x >> 27)) * 0x94D049BB133111EB
	return x ^ (x >> 31)
}

func noise(nSamples int, seed uint64) []byte {
	iq := make([]byte, 2*nSamples)
	for n := 0; n < nSamples; n++ {
		h := splitmix64(seed ^ uint64(n))
		iq[2*n] = byte(119 + bits.OnesCount16(uint16(h)))
		iq[2*n+1] = byte(120 + bits.OnesCount16(uint16(h>>16)))
	}
	return iq
}

type packet struct {
	start  int
	data   []byte
	nBits  int
	dI, dQ int
}

func packetSchedule(nPackets, nSamples, packetSamples int, seed uint64, edgeEvery, blockSize int) []int {
	stride := nSamples / nPackets
	if stride <= packetSamples+64 {
		log.Fatal("packets would overlap")
	}
	start := make([]int, nPackets)
	for j := range start {
		jitter := splitmix64(seed^(uint64(j)+0xABCDEF)) % uint64(stride-packetSamples-32)
		start[j] = j*stride + int(jitter)
	}
	if edgeEvery > 0 && blockSize > 0 {
		for i := 0; i < nPackets; i += edgeEvery {
			b := (start[i] + packetSamples/2 + blockSize - 1) / blockSize * blockSize
			s := b - packetSamples/2
			if s >= i*stride && s+packetSamples < (i+1)*stride {
				start[i] = s
			}
		}
	}
	return start
}

func clampAdd(v byte, d int) byte {
	x := int(v) + d
	if x < 0 {
		x = 0
	}
	if x > 255 {
		x = 255
	}
	return byte(x)
}

// plant: Manchester-OOK, bit 1 = chip high then low, bit 0 = low then high; "high" adds (dI, dQ) with clamping
func plant(iq []byte, pk []packet, chip int) {
	nSamples := len(iq) / 2
	sl := 2 * chip
	for _, p := range pk {
		for b := 0; b < p.nBits; b++ {
			bit := (p.data[b>>3] >> (7 - uint(b&7))) & 1
			for w := 0; w < sl; w++ {
				if (w < chip) != (bit == 1) {
					continue
				}
				n := p.start + b*sl + w
				if n < 0 || n >= nSamples {
					continue
				}
				iq[2*n] = clampAdd(iq[2*n], p.dI)
				iq[2*n+1] = clampAdd(iq[2*n+1], p.dQ)
			}
		}
	}
}

var (
	bch   = crc.NewCRC("BCH", 0, 0x6F63, 0)
	ccitt = crc.NewCRC("CCITT", 0xFFFF, 0x1021, 0x1D0F)
)

func be(v uint64, n int) []byte {
	b := make([]byte, n)
	for i := n - 1; i >= 0; i-- {
		b[i] = byte(v)
		v >>= 8
	}
	return b
}

// buildSCM: the inverse of scm.NewSCM (scm/scm.go:103-109), BCH residue 0 over bytes 2..11
func buildSCM(id, typ, consumption uint64) []byte {
	var v uint64 = 0x1F2A60 // 21-bit preamble 111110010101001100000
	v = v<<2 | (id>>24)&3
	v = v<<1 | 0
	v = v<<2 | 0          // physical tamper
	v = v<<4 | typ&15
	v = v<<2 | 0          // encoder tamper
	// 32 bits so far; consumption (24) and id (24) follow: 80 bits = 10 bytes
	body := append(be(v, 4), be(consumption&0xFFFFFF, 3)...)
	body = append(body, be(id&0xFFFFFF, 3)...)
	c := bch.Checksum(body[2:])
	return append(body, byte(c>>8), byte(c))
}

func buildIDM(serial, consumption uint64) []byte {
	B := make([]byte, 92)
	for i := range B {
		B[i] = 0x5A
	}
	copy(B[0:4], []byte{0x55, 0x55, 0x16, 0xA3})
	B[4], B[5], B[6], B[7], B[8] = 0x1C, 0x5C, 0xC6, 0x04, 7
	copy(B[9:13], be(serial, 4))
	copy(B[29:33], be(consumption, 4))
	c := ccitt.Checksum(B[9:13]) ^ 0xFFFF
	B[88], B[89] = byte(c>>8), byte(c)
	c = ccitt.Checksum(B[4:90]) ^ 0xFFFF
	B[90], B[91] = byte(c>>8), byte(c)
	return B
}

func buildSCMPlus(id, consumption uint64) []byte {
	B := make([]byte, 16)
	B[0], B[1], B[2], B[3] = 0x16, 0xA3, 0x1E, 0x9C
	copy(B[4:8], be(id, 4))
	copy(B[8:12], be(consumption, 4))
	B[12], B[13] = 0x02, 0x48
	c := ccitt.Checksum(B[2:14]) ^ 0xFFFF
	B[14], B[15] = byte(c>>8), byte(c)
	return B
}

// builder: tests/util.py PKT_BUILDERS
func builder(kind string, i int) ([]byte, int) {
	u := uint64(i)
	switch kind {
	case "scm":
		return buildSCM(1000+u*7919, u%12+1, (u*104729)&0xFFFFFF), 96
	case "idm":
		return buildIDM(2000+u*7919, u*31), 736
	case "netidm":
		return buildIDM(3000+u*7919, u*17), 736
	case "scm+":
		return buildSCMPlus(4000+u*7919, u*13), 128
	}
	return nil, 0
}

// synthStream: tests/util.py synth_stream (edge_every 4, amplitudes (30, -26))
func synthStream(protos []string, chip, nBlocks, blockSize int, seed uint64, nPackets int) []byte {
	nSamples := nBlocks * blockSize
	iq := noise(nSamples, seed)
	var kinds []string
	longest := 0
	for _, p := range protos {
		if _, n := builder(p, 0); n > 0 {
			kinds = append(kinds, p)
			if n*2*chip > longest {
				longest = n * 2 * chip
			}
		}
	}
	if nPackets == 0 || len(kinds) == 0 {
		return iq
	}
	starts := packetSchedule(nPackets, nSamples, longest, seed, 4, blockSize)
	pk := make([]packet, nPackets)
	for i, s := range starts {
		data, n := builder(kinds[i%len(kinds)], i)
		sign := -1
		if i%2 == 1 {
			sign = 1
		}
		pk[i] = packet{s, data, n, sign * 30, -sign*(-26) + i%5}
	}
	plant(iq, pk, chip)
	return iq
}

// ---------------------------------------------------------------- golden files

type synthCase struct {
	Name      string   `json:"name"`
	Protocols []string `json:"protocols"`
	Chip      int      `json:"chip"`
	Blocks    int      `json:"blocks"`
	Seed      uint64   `json:"seed"`
	Packets   int      `json:"packets"`
	IqSha     string   `json:"iq_sha"`
	QSha      string   `json:"qsha"`
	NHits     int      `json:"n_hits"`
	HitsSha   string   `json:"hits_sha"`
	PktSha    string   `json:"pkt_sha"`
}

type sampleCase struct {
	Name      string   `json:"name"`
	Protocols []string `json:"protocols"`
	Chip      int      `json:"chip"`
	Blocks    int      `json:"blocks"`
	BlockSize int      `json:"block_size"`
	Ones      int      `json:"ones"`
	QSha      string   `json:"qsha"`
	Hits      [][2]int `json:"hits"`
	PktSha    string   `json:"pkt_sha"`
}

func writeJSON(path string, v interface{}) {
	b, err := json.MarshalIndent(v, "", " ")
	if err != nil {
		log.Fatal(err)
	}
	if err := os.WriteFile(path, append(b, '\n'), 0o644); err != nil {
		log.Fatal(err)
	}
	log.Println("wrote", path)
}

func main() {
	repo := flag.String("repo", "", "checkout of the MI355X repository (tests/golden/ is read and written there)")
	sample := flag.String("sample", "assets/sample.bin", "the reference capture (relative to the rtlamr checkout)")
	iqdir := flag.String("iqdir", "", "also write every synthetic stream as <iqdir>/<case>.iq (input of protocol/decode_amd_test.go)")
	flag.Parse()
	if *repo == "" {
		log.Fatal("-repo is required")
	}
	golden := filepath.Join(*repo, "tests", "golden")

	// ---- synthetic streams: the cases of tests/golden/synth.json, same seeds ----
	var have struct {
		Cases []synthCase `json:"cases"`
	}
	raw, err := os.ReadFile(filepath.Join(golden, "synth.json"))
	if err != nil {
		log.Fatal(err)
	}
	if err := json.Unmarshal(raw, &have); err != nil {
		log.Fatal(err)
	}
	var out struct {
		Generator string      `json:"generator"`
		Cases     []synthCase `json:"cases"`
	}
	out.Generator = "go/cmd/amdgolden over github.com/bemasher/rtlamr/protocol (reference Go implementation)"
	for _, c := range have.Cases {
		probe := newRefDecoder(c.Protocols, c.Chip) // geometry only
		iq := synthStream(c.Protocols, c.Chip, c.Blocks, probe.d.Cfg.BlockSize, c.Seed, c.Packets)
		if got := shaHex(iq); got != c.IqSha {
			log.Fatalf("%s: the Go port of the synthetic stream differs from the recorded one (iq sha %s, want %s)", c.Name, got, c.IqSha)
		}
		if *iqdir != "" {
			if err := os.WriteFile(filepath.Join(*iqdir, c.Name+".iq"), iq, 0o644); err != nil {
				log.Fatal(err)
			}
		}
		res := decodeStream(c.Protocols, c.Chip, iq)
		g := c
		g.QSha = shaHex(res.q)
		g.NHits = len(res.hits)
		g.HitsSha = hitsSha(res.hits)
		g.PktSha = pktSha(res.hits, true, res.cfg.PacketSymbols/8)
		out.Cases = append(out.Cases, g)
		fmt.Printf("%-12s hits %6d  q %s\n", c.Name, g.NHits, g.QSha[:16])
	}
	writeJSON(filepath.Join(golden, "go_synth.json"), out)

	// ---- the reference capture, the cases of tests/golden/sample_bin.json ----
	capture, err := os.ReadFile(*sample)
	if err != nil {
		log.Fatal(err)
	}
	lut := protocol.NewMagLUT()
	lutBytes := make([]byte, 0, 1024)
	for _, v := range lut {
		lutBytes = binary.LittleEndian.AppendUint32(lutBytes, math.Float32bits(v))
	}
	var sb struct {
		Generator string       `json:"generator"`
		FileSha   string       `json:"file_sha256"`
		FileBytes int          `json:"file_bytes"`
		LutSha    string       `json:"lut_sha256"`
		Cases     []sampleCase `json:"cases"`
	}
	sb.Generator = out.Generator
	sb.FileSha, sb.FileBytes, sb.LutSha = shaHex(capture), len(capture), shaHex(lutBytes)
	for _, c := range []struct {
		name   string
		protos []string
		chip   int
		nbytes int
	}{
		{"cfg1_first_512KiB_scm72", []string{"scm"}, 72, 524288},
		{"whole_scm72", []string{"scm"}, 72, 0},
		{"whole_idm72", []string{"idm"}, 72, 0},
		{"whole_scm80", []string{"scm"}, 80, 0},
	} {
		n := c.nbytes
		if n == 0 {
			n = len(capture)
		}
		res := decodeStream(c.protos, c.chip, capture[:n])
		sc := sampleCase{Name: c.name, Protocols: c.protos, Chip: c.chip, Blocks: n / res.cfg.BlockSize2,
			BlockSize: res.cfg.BlockSize, Ones: ones(res.q), QSha: shaHex(res.q), Hits: [][2]int{}}
		for _, h := range res.hits { // decode order: call ascending, idx ascending
			sc.Hits = append(sc.Hits, [2]int{h.call, h.idx})
		}
		sc.PktSha = pktSha(res.hits, false, (res.cfg.PacketSymbols+7)/8)
		sb.Cases = append(sb.Cases, sc)
		fmt.Printf("%-24s hits %4d  q %s\n", c.name, len(sc.Hits), sc.QSha[:16])
	}
	writeJSON(filepath.Join(golden, "go_sample_bin.json"), sb)

	// ---- the r900 parser's second stage (r900/r900.go:82-150), only with -tags amdgolden (r900_golden.go) ----
	r900Golden(golden, capture)
}
