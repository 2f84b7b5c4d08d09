//go:build amd

// Package protocol: MI355X build of the decoder (build tag "amd").
//
// This file takes the place of protocol/decode.go (give that file the
// constraint "//go:build !amd"); parse.go and every parser package stay as they
// are.  The exported surface is the one of decode.go -- PacketConfig, Decoder
// with its Cfg / Signal / Quantized fields, NewDecoder, Log, RegisterProtocol,
// Allocate, Decode, Filter, Search, Slice, Demodulator, MagLUT, NewMagLUT,
// NextPowerOf2 -- and the
// arithmetic of Decode (decode.go:163-172 and :255-375: history slide, magnitude
// LUT, cumulative-sum matched filter, quantize, preamble Search, Slice) runs in
// libamrdemod.so on the GPU, bit for bit (include/amrdemod.h).
//
//	go build -tags amd ./...      with CGO_CFLAGS=-I<repo>/include  CGO_LDFLAGS="-L<repo>/rtlamr_amd/csrc -lamrdemod"
//
// Decode accepts one block, as main.go:235 passes it, or any whole number of
// blocks: the library then processes them as one batch, and the parsers still see
// them call by call, in order.
package protocol

/*
#cgo LDFLAGS: -lamrdemod
#include <stdlib.h>
#include "amrdemod.h"
*/
import "C"

import (
	"fmt"
	"log"
	"math"
	"os"
	"strconv"
	"strings"
	"sync"
	"unsafe"
)

// PacketConfig specifies packet-specific radio configuration (decode.go:27-42).
type PacketConfig struct {
	Protocol string
	Preamble string

	DataRate int

	BlockSize, BlockSize2    int
	ChipLength, SymbolLength int
	SampleRate               int

	PreambleSymbols, PacketSymbols int
	PreambleLength, PacketLength   int

	BufferLength int
	CenterFreq   uint32
}

// Decoder contains the radio configuration and the handle of the GPU decoder.
// Signal and Quantized keep their reference meaning for parsers that read them
// (r900/r900.go:162-170 reads Signal[SymbolLength:] inside Parse): Signal is
// refilled per call on the host when such a parser is registered; Quantized is
// read by no parser and is filled only on request (KeepQuantized).
type Decoder struct {
	Cfg PacketConfig
	wg  *sync.WaitGroup

	Signal    []float32
	Quantized []byte

	// KeepQuantized makes Decode copy the batch's bit decisions back into
	// Quantized[PacketLength:] after every call (tests, tools, and the exported
	// Search / Slice helpers below, which read Quantized); off by default.
	KeepQuantized bool
	// KeepSignal makes Decode refill Signal on the host even when no registered
	// parser needs it (the exported Filter helper reads a caller's signal anyway).
	KeepSignal bool

	// Device is the HIP device ordinal Allocate opens.  -1 (NewDecoder's default)
	// takes it from the environment: AMR_DEVICE, else LOCAL_RANK (one process per
	// GPU, the deployment amr_comm_* is made for), else 0.
	Device int
	// Devices, when it names more than one HIP device, makes this ONE Decoder drive all of them from this one process
	// (the reference caller is one process, main.go:59-128): Allocate opens a handle per device and joins them in one
	// RCCL communicator (amr_comm_init_all: the grouped form a single thread needs), and every Decode call's blocks are
	// cut into len(Devices) contiguous ranges, range r decoded on Devices[r] after priming it with the blocks in front of
	// the range (amr_prime) -- the split SURVEY.md 8e describes, results identical to one device's (except the never-cleared high bits of a range's first packet's last byte when PacketSymbols%8 != 0: decodeOnDevices).  Empty: Device alone.  (KeepQuantized reads back handle 0's range only: a
	// diagnostic, use one device for it.)
	Devices []int

	st *amdState // shared by the copies Decode's value receiver makes
}

type amdState struct {
	h          *C.amr_handle       // the only handle, or the one on Devices[0]
	hs         []*C.amr_handle     // Devices: one handle per device, hs[0] == h (nil for a single device)
	tail       []byte              // Devices: the last prime-blocks + halo bytes of the stream so far (history of range 0)
	pin        unsafe.Pointer      // Devices: pinned staging of tail ++ input (amr_host_alloc)
	pinCap     int
	parsers    []Parser            // in registration order
	preambles  map[string][]Parser // key: Cfg().Preamble
	order      []string            // distinct preambles in registration order
	pid        map[string]int      // preamble -> id used by the library
	protocols  []string
	needSignal bool
	lut        MagLUT
	calls      uint64 // Decode calls (blocks) made so far = call index of the next block
	csum       []float32 // scratch of the exported Filter helper
	pkt        []byte    // scratch of the exported Slice helper (never cleared, like the reference's d.pkt)
}

func NewDecoder() Decoder {
	return Decoder{
		wg:     new(sync.WaitGroup),
		Device: -1,
		st:     &amdState{preambles: make(map[string][]Parser), pid: make(map[string]int)},
	}
}

func deviceFromEnv() int {
	for _, name := range []string{"AMR_DEVICE", "LOCAL_RANK"} {
		if v, err := strconv.Atoi(os.Getenv(name)); err == nil && v >= 0 {
			return v
		}
	}
	return 0
}

func (d Decoder) Log() {
	log.Println("CenterFreq:", d.Cfg.CenterFreq)
	log.Println("SampleRate:", d.Cfg.SampleRate)
	log.Println("DataRate:", d.Cfg.DataRate)
	log.Println("ChipLength:", d.Cfg.ChipLength)
	log.Println("PreambleSymbols:", d.Cfg.PreambleSymbols)
	log.Println("PreambleLength:", d.Cfg.PreambleLength)
	log.Println("PacketSymbols:", d.Cfg.PacketSymbols)
	log.Println("PacketLength:", d.Cfg.PacketLength)
	log.Println("Protocols:", strings.Join(d.st.protocols, ","))
	log.Println("Preambles:", strings.Join(d.st.order, ","))
	if d.st.h != nil {
		buf := make([]byte, 512)
		C.amr_describe(d.st.h, (*C.char)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)))
		log.Println("Device:", C.GoString((*C.char)(unsafe.Pointer(&buf[0]))))
	}
}

// RegisterProtocol: decode.go:100-128.  The field-wise maxima are kept here for
// Log and for callers that read Cfg before Allocate (main.go:94); the library
// recomputes them from the same inputs in Allocate.
func (d *Decoder) RegisterProtocol(p Parser) {
	p.SetDecoder(d)
	c := p.Cfg()
	d.Cfg.CenterFreq = c.CenterFreq
	d.Cfg.DataRate = max(d.Cfg.DataRate, c.DataRate)
	d.Cfg.ChipLength = max(d.Cfg.ChipLength, c.ChipLength)
	d.Cfg.PreambleSymbols = max(d.Cfg.PreambleSymbols, c.PreambleSymbols)
	d.Cfg.PacketSymbols = max(d.Cfg.PacketSymbols, c.PacketSymbols)

	s := d.st
	if _, known := s.preambles[c.Preamble]; !known {
		s.order = append(s.order, c.Preamble)
	}
	s.preambles[c.Preamble] = append(s.preambles[c.Preamble], p)
	s.parsers = append(s.parsers, p)
	s.protocols = append(s.protocols, c.Protocol)
	// r900 (and r900bcd, which wraps it and reports Protocol "r900") reads d.Signal in Parse
	if c.Protocol == "r900" {
		s.needSignal = true
	}
}

// DeviceError is what a failed library call panics with.  The reference's Decoder API has no error returns (its only
// failure, a short block, is a runtime panic: decode.go:222), so the binding keeps the signatures and panics too -- with
// a typed value a library caller can recover() and inspect, instead of ending the process (main.go, a CLI, simply lets
// it propagate: same exit as log.Fatalf, with a stack).
type DeviceError struct {
	Call   string // the C entry point
	Status int    // amr_status (include/amrdemod.h): AMR_EINVAL -1, AMR_ENOMEM -2, AMR_EHIP -3, AMR_ENODEV -4, AMR_EOVERFLOW -5
	Text   string // amr_strerror + amr_last_error
}

func (e *DeviceError) Error() string { return fmt.Sprintf("%s: status %d: %s", e.Call, e.Status, e.Text) }

func fatal(what string, st C.amr_status) {
	panic(&DeviceError{Call: what, Status: int(st),
		Text: C.GoString(C.amr_strerror(st)) + ": " + C.GoString(C.amr_last_error())})
}

// Allocate: decode.go:131-160.  The geometry comes back from the library, which
// computes it exactly as the reference does (amr_plan / amr_create).
func (d *Decoder) Allocate() {
	s := d.st
	protos := make([]C.amr_protocol, len(s.parsers))
	for i, p := range s.parsers {
		c := p.Cfg()
		cs := C.CString(c.Preamble) // C memory; read by the library during amr_create only
		defer C.free(unsafe.Pointer(cs))
		protos[i] = C.amr_protocol{
			preamble:         cs,
			data_rate:        C.int32_t(c.DataRate),
			chip_length:      C.int32_t(c.ChipLength),
			preamble_symbols: C.int32_t(c.PreambleSymbols),
			packet_symbols:   C.int32_t(c.PacketSymbols),
		}
	}
	dev := d.Device
	if dev < 0 {
		dev = deviceFromEnv()
	}
	if len(d.Devices) > 1 {
		// one handle per device, then ONE grouped communicator init for all of them (a loop of amr_comm_init from this
		// single thread would wait in the first ncclCommInitRank for ever)
		ords := make([]C.int32_t, len(d.Devices))
		for i, v := range d.Devices {
			ords[i] = C.int32_t(v)
		}
		if st := C.amr_comm_check_all(nil, &ords[0], C.int32_t(len(ords)), 0, 1<<16); st != C.AMR_OK {
			fatal("amr_comm_check_all", st)
		}
		s.hs = make([]*C.amr_handle, len(d.Devices))
		for i := range d.Devices {
			if st := C.amr_create(&protos[0], C.int32_t(len(protos)), ords[i], &s.hs[i]); st != C.AMR_OK {
				fatal("amr_create", st)
			}
		}
		if st := C.amr_comm_init_all(&s.hs[0], C.int32_t(len(s.hs)), 0, 1<<16); st != C.AMR_OK {
			fatal("amr_comm_init_all", st)
		}
		s.h = s.hs[0]
	} else {
		if len(d.Devices) == 1 {
			dev = d.Devices[0]
		}
		if st := C.amr_create(&protos[0], C.int32_t(len(protos)), C.int32_t(dev), &s.h); st != C.AMR_OK {
			fatal("amr_create", st)
		}
	}
	var g C.amr_geometry
	if st := C.amr_get_geometry(s.h, &g); st != C.AMR_OK {
		fatal("amr_get_geometry", st)
	}
	d.Cfg.SymbolLength, d.Cfg.SampleRate = int(g.symbol_length), int(g.sample_rate)
	d.Cfg.PreambleLength, d.Cfg.PacketLength = int(g.preamble_length), int(g.packet_length)
	d.Cfg.BlockSize, d.Cfg.BlockSize2 = int(g.block_size), int(g.block_size2)
	d.Cfg.BufferLength = int(g.buffer_length)
	for i, p := range s.parsers {
		s.pid[p.Cfg().Preamble] = int(C.amr_preamble_id(s.h, C.int32_t(i)))
	}
	d.Signal = make([]float32, d.Cfg.BlockSize+d.Cfg.SymbolLength)
	d.Quantized = make([]byte, d.Cfg.BufferLength)
	s.lut = NewMagLUT()
	s.csum = make([]float32, len(d.Signal)+1)
	s.pkt = make([]byte, (d.Cfg.PacketSymbols+7)>>3)
}

// Decode accepts a sample block (or several) and returns a channel of messages,
// closed when every parser has finished (decode.go:163-197).
func (d Decoder) Decode(input []byte) chan Message {
	s := d.st
	bs, bs2 := d.Cfg.BlockSize, d.Cfg.BlockSize2
	nBlocks := len(input) / bs2
	if nBlocks == 0 {
		panic("runtime error: index out of range") // what decode.go:222 does with a short block
	}
	first := s.calls // call index of input's first block
	np := len(s.order)
	type perCall [][]Data // [preamble id][]Data
	calls := make([]perCall, nBlocks)
	for k := range calls {
		calls[k] = make(perCall, np)
	}
	// The result arrays belong to the handle until the next amr_* call on it: turn them
	// into []Data at once (NewData copies the bytes, parse.go:61-69).  Hits of a
	// preamble are sorted by (call, idx), so one pass per preamble splits them by call.
	take := func(res *C.amr_result) {
		n, pb := int(res.n_hits), int(res.pkt_bytes)
		if n == 0 {
			return
		}
		off := unsafe.Slice((*uint64)(unsafe.Pointer(res.preamble_offset)), int(res.n_preambles)+1)
		blk := unsafe.Slice((*uint64)(unsafe.Pointer(res.hit_block)), n)
		idx := unsafe.Slice((*uint32)(unsafe.Pointer(res.hit_idx)), n)
		pkt := unsafe.Slice((*byte)(unsafe.Pointer(res.pkt)), n*pb)
		for p := 0; p < int(res.n_preambles); p++ {
			for i := int(off[p]); i < int(off[p+1]); i++ {
				data := NewData(pkt[i*pb : (i+1)*pb])
				data.Idx = int(idx[i])
				k := int(blk[i] - first)
				calls[k][p] = append(calls[k][p], data)
			}
		}
	}
	if len(s.hs) > 1 {
		d.decodeOnDevices(input, nBlocks, first, take)
	} else {
		var res C.amr_result
		// input is Go memory: the library reads it during the call only (cgo pointer rules)
		if st := C.amr_decode_batch(s.h, (*C.uint8_t)(unsafe.Pointer(&input[0])), C.size_t(nBlocks*bs2),
			C.size_t(nBlocks), &res); st != C.AMR_OK {
			fatal("amr_decode_batch", st)
		}
		take(&res) // res.first_block == first: amr_decode_batch never defers
	}
	s.calls += uint64(nBlocks)
	var q []byte
	if d.KeepQuantized {
		q = make([]byte, nBlocks*bs/8)
		if st := C.amr_copy_quantized(s.h, (*C.uint8_t)(unsafe.Pointer(&q[0])), C.size_t(len(q))); st != C.AMR_OK {
			fatal("amr_copy_quantized", st)
		}
	}

	msgCh := make(chan Message)
	go func() {
		for k := 0; k < nBlocks; k++ {
			if s.needSignal || d.KeepSignal { // MagLUT.Execute on the host, same table: bit-identical to decode.go:165,169
				copy(d.Signal, d.Signal[bs:])
				s.lut.Execute(input[k*bs2:(k+1)*bs2], d.Signal[d.Cfg.SymbolLength:])
			}
			if d.KeepQuantized {
				copy(d.Quantized, d.Quantized[bs:]) // decode.go:166
				fresh := d.Quantized[d.Cfg.PacketLength:]
				for i := 0; i < bs; i++ {
					fresh[i] = (q[(k*bs+i)>>3] >> (7 - uint(i&7))) & 1
				}
			}
			for preamble, parsers := range s.preambles { // decode.go:177
				pkts := calls[k][s.pid[preamble]]
				d.wg.Add(len(parsers))
				for _, p := range parsers {
					go p.Parse(pkts, msgCh, d.wg) // decode.go:185-187
				}
			}
			// the next call may not start before this call's parsers are done (main.go drains
			// the channel to its close before calling Decode again; r900 keeps per-call state)
			d.wg.Wait()
		}
		close(msgCh) // decode.go:191-194
	}()
	return msgCh
}

// decodeOnDevices: one Decode call's blocks over the handles of Devices.  Range r = blocks [k0, k1) goes to hs[r], which
// first forgets its state (amr_reset) and is primed with the amr_prime_blocks blocks in front of k0 -- out of this call's
// input, or for the first ranges out of the tail kept from the previous calls -- so that its histories are the single
// Decoder's (decode.go:165-166); amr_set_block_base makes its call indices the stream's.  Every range is submitted before
// any is collected: the devices run side by side, driven by this one thread.  The hit records come back per handle
// (host memory, `take`); a host that wants them merged on one device instead -- a replay loop that consumes validated
// records one step behind, like bench.py -- posts amr_gather_hits_all on the same handles (INTEGRATION.md).
//
// Cost and limits.  Every call re-primes every device that has blocks (amr_prime_blocks blocks each): Devices is for
// callers that hand Decode MANY blocks per call (nBlocks >> len(Devices) * amr_prime_blocks); with the unchanged
// one-block-per-call loop of main.go:235 use one device.  One thing differs from a single device: amr_reset zeroes the
// byte Decoder.Slice never clears (decode.go:363-366), so with PacketSymbols%8 != 0 (r900 alone or with scm) the high bits
// of the LAST byte of each range's first hit start from zero instead of continuing the previous hit's -- parsers never
// read those bits; a caller that needs them patches with the rule of amr_set_stale_carry (include/amrdemod.h).
func (d Decoder) decodeOnDevices(input []byte, nBlocks int, first uint64, take func(*C.amr_result)) {
	s := d.st
	bs2 := d.Cfg.BlockSize2
	n := len(s.hs)
	pb := int(C.amr_prime_blocks(s.h))
	halo := int(C.amr_halo_bytes(s.h))
	// the stream so far, as far back as any range's priming can reach: tail ++ input -- in pinned C memory
	// (amr_host_alloc): amr_submit_host goes on reading its input after it has returned, which the cgo rules allow for C
	// memory only, and pinned memory makes the transfer a true DMA
	need := len(s.tail) + nBlocks*bs2
	if need > s.pinCap {
		if s.pin != nil {
			C.amr_host_free(s.pin)
		}
		if st := C.amr_host_alloc(C.size_t(need), &s.pin); st != C.AMR_OK {
			fatal("amr_host_alloc", st)
		}
		s.pinCap = need
	}
	hist := unsafe.Slice((*byte)(s.pin), need)
	copy(hist, s.tail)
	copy(hist[len(s.tail):], input[:nBlocks*bs2])
	base := len(s.tail) // offset of input's first block inside hist
	ranges := make([][2]int, n)
	for r := 0; r < n; r++ {
		q, rem := nBlocks/n, nBlocks%n
		k0 := r*q + min(r, rem)
		k1 := k0 + q
		if r < rem {
			k1++
		}
		ranges[r] = [2]int{k0, k1}
	}
	for r, h := range s.hs {
		k0, k1 := ranges[r][0], ranges[r][1]
		if k1 == k0 { // fewer blocks than devices: nothing to reset, prime or submit on this one (ADVICE r05)
			continue
		}
		if st := C.amr_reset(h); st != C.AMR_OK {
			fatal("amr_reset", st)
		}
		start := base + k0*bs2                  // first byte of the range inside hist
		p0 := max(base%bs2, start-pb*bs2)       // whole blocks only, never before the stream's start
		if p0 < start {
			var lead *C.uint8_t
			if p0 >= halo {
				lead = (*C.uint8_t)(unsafe.Pointer(&hist[p0-halo]))
			}
			if st := C.amr_prime(h, lead, (*C.uint8_t)(unsafe.Pointer(&hist[p0])), C.size_t((start-p0)/bs2), 0); st != C.AMR_OK {
				fatal("amr_prime", st)
			}
		}
		if st := C.amr_set_block_base(h, C.uint64_t(first+uint64(k0))); st != C.AMR_OK {
			fatal("amr_set_block_base", st)
		}
		if k1 > k0 {
			if st := C.amr_submit_host(h, (*C.uint8_t)(unsafe.Pointer(&hist[start])), C.size_t((k1-k0)*bs2), C.size_t(k1-k0)); st != C.AMR_OK {
				fatal("amr_submit_host", st)
			}
		}
	}
	for r, h := range s.hs {
		if ranges[r][1] == ranges[r][0] {
			continue
		}
		var res C.amr_result
		if st := C.amr_collect(h, &res); st != C.AMR_OK {
			fatal("amr_collect", st)
		}
		take(&res)
	}
	keep := pb*bs2 + halo
	if len(hist) > keep {
		hist = hist[len(hist)-keep:]
	}
	s.tail = append(s.tail[:0], hist...)
}

// Close releases the GPU decoder (no counterpart in the reference, whose buffers are garbage collected).
func (d *Decoder) Close() {
	if d.st == nil {
		return
	}
	for _, h := range d.st.hs[min(1, len(d.st.hs)):] {
		C.amr_destroy(h) // destroys its communicator rank first
	}
	d.st.hs = nil
	if d.st.pin != nil {
		C.amr_host_free(d.st.pin)
		d.st.pin, d.st.pinCap = nil, 0
	}
	if d.st.h != nil {
		C.amr_destroy(d.st.h)
		d.st.h = nil
	}
}

// Filter, Search and Slice are exported by the reference (decode.go:229, 255, 353) although only Decode calls them.
// They stay available as HOST-side diagnostics with the reference's meaning, for third-party callers that compile
// against them; the hot path does not go through them (it runs in the library).  Search and Slice read d.Quantized,
// i.e. they need KeepQuantized; Filter works on the slices it is given.

// Filter: running float32 sum restarted at zero, then the sign of (lower chip) - (upper chip) per output sample.
func (d Decoder) Filter(input []float32, output []byte) {
	c := d.st.csum
	if len(c) < len(input)+1 {
		c = make([]float32, len(input)+1)
	}
	c[0] = 0
	var acc float32
	for i := range input {
		acc += input[i]
		c[i+1] = acc
	}
	cl, sl := d.Cfg.ChipLength, d.Cfg.SymbolLength
	for i := range output {
		mid := c[i+cl]
		f := (mid - c[i]) - (c[i+sl] - mid)
		output[i] = 1 - byte(math.Float32bits(f)>>31)
	}
}

// Search: every index in [0, BlockSize) at which all preamble bits match Quantized at a stride of SymbolLength,
// ascending.  For every legal -symbollength this is exactly the set the reference's byte-prefiltered two-pass search
// returns (SURVEY.md 8a; tests/test_oracle_golden.py compares the two forms).
func (d *Decoder) Search(preamble []byte) []int {
	var hits []int
	sl := d.Cfg.SymbolLength
	for idx := 0; idx < d.Cfg.BlockSize; idx++ {
		ok := true
		for p, bit := range preamble {
			if d.Quantized[idx+p*sl] != bit {
				ok = false
				break
			}
		}
		if ok {
			hits = append(hits, idx)
		}
	}
	return hits
}

// Slice: for every index, PacketSymbols decisions at a stride of SymbolLength shifted MSB-first into bytes.
func (d Decoder) Slice(indices []int) (pkts []Data) {
	pkt := d.st.pkt
	for _, q := range indices {
		if q > d.Cfg.BlockSize {
			continue
		}
		for p := 0; p < d.Cfg.PacketSymbols; p++ {
			pkt[p>>3] = pkt[p>>3]<<1 | d.Quantized[q+p*d.Cfg.SymbolLength]
		}
		data := NewData(pkt)
		data.Idx = q
		pkts = append(pkts, data)
	}
	return
}

// A Demodulator knows how to demodulate an array of uint8 IQ samples into an
// array of float32 samples (decode.go:198-202).
type Demodulator interface {
	Execute([]byte, []float32)
}

// MagLUT is the magnitude lookup table (decode.go:205-225); the library builds the
// same table for the GPU (amr_get_mag_lut returns it).
type MagLUT []float32

func NewMagLUT() (lut MagLUT) {
	lut = make([]float32, 0x100)
	for idx := range lut {
		lut[idx] = (127.5 - float32(idx)) / 127.5
		lut[idx] *= lut[idx]
	}
	return
}

func (lut MagLUT) Execute(input []byte, output []float32) {
	i := 0
	for idx := range output {
		output[idx] = lut[input[i]] + lut[input[i+1]]
		i += 2
	}
}

func NextPowerOf2(v int) int {
	return 1 << uint(math.Ceil(math.Log2(float64(v))))
}
