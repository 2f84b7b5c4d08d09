//go:build amd

// Package protocol: MI355X build of the decoder (build tag "amd").
//
// This file takes the place of protocol/decode.go (give that file the
// constraint "//go:build !amd"); parse.go and every parser package stay as they
// are.  The exported surface is the one of decode.go -- PacketConfig, Decoder
// with its Cfg / Signal / Quantized fields, NewDecoder, Log, RegisterProtocol,
// Allocate, Decode, Demodulator, MagLUT, NewMagLUT, NextPowerOf2 -- and the
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
	"log"
	"math"
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
	// Quantized[PacketLength:] after every call (tests, tools); off by default.
	KeepQuantized bool

	st *amdState // shared by the copies Decode's value receiver makes
}

type amdState struct {
	h          *C.amr_handle
	parsers    []Parser            // in registration order
	preambles  map[string][]Parser // key: Cfg().Preamble
	order      []string            // distinct preambles in registration order
	pid        map[string]int      // preamble -> id used by the library
	protocols  []string
	needSignal bool
	lut        MagLUT
	calls      uint64 // Decode calls (blocks) made so far = call index of the next block
}

func NewDecoder() Decoder {
	return Decoder{
		wg: new(sync.WaitGroup),
		st: &amdState{preambles: make(map[string][]Parser), pid: make(map[string]int)},
	}
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

func max(a, b int) int {
	if a > b {
		return a
	}
	return b
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

func fatal(what string, st C.amr_status) {
	log.Fatalf("%s: %s: %s", what, C.GoString(C.amr_strerror(st)), C.GoString(C.amr_last_error()))
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
	if st := C.amr_create(&protos[0], C.int32_t(len(protos)), 0, &s.h); st != C.AMR_OK {
		fatal("amr_create", st)
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
	var res C.amr_result
	// input is Go memory: the library reads it during the call only (cgo pointer rules)
	if st := C.amr_decode_batch(s.h, (*C.uint8_t)(unsafe.Pointer(&input[0])), C.size_t(nBlocks*bs2),
		C.size_t(nBlocks), &res); st != C.AMR_OK {
		fatal("amr_decode_batch", st)
	}
	n := int(res.n_hits)
	np := int(res.n_preambles)
	pb := int(res.pkt_bytes)
	off := unsafe.Slice((*uint64)(unsafe.Pointer(res.preamble_offset)), np+1)
	var blk []uint64
	var idx []uint32
	var pkt []byte
	if n > 0 {
		blk = unsafe.Slice((*uint64)(unsafe.Pointer(res.hit_block)), n)
		idx = unsafe.Slice((*uint32)(unsafe.Pointer(res.hit_idx)), n)
		pkt = unsafe.Slice((*byte)(unsafe.Pointer(res.pkt)), n*pb)
	}
	first := s.calls
	s.calls += uint64(nBlocks)

	// The result arrays belong to the handle until the next amr_* call: turn them
	// into []Data now (NewData copies the bytes, parse.go:61-69).  Hits of a
	// preamble are sorted by (call, idx), so one pass per preamble splits them by call.
	type perCall [][]Data // [preamble id][]Data
	calls := make([]perCall, nBlocks)
	for k := range calls {
		calls[k] = make(perCall, np)
	}
	for p := 0; p < np; p++ {
		for i := int(off[p]); i < int(off[p+1]); i++ {
			k := int(blk[i] - first)
			data := NewData(pkt[i*pb : (i+1)*pb])
			data.Idx = int(idx[i])
			calls[k][p] = append(calls[k][p], data)
		}
	}
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
			if s.needSignal || d.KeepQuantized {
				copy(d.Signal, d.Signal[bs:]) // decode.go:165
			}
			if s.needSignal { // MagLUT.Execute on the host, same table: bit-identical to decode.go:169
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

// Close releases the GPU decoder (no counterpart in the reference, whose buffers are garbage collected).
func (d *Decoder) Close() {
	if d.st != nil && d.st.h != nil {
		C.amr_destroy(d.st.h)
		d.st.h = nil
	}
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
