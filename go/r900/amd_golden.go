//go:build amdgolden

// Test hook for go/cmd/amdgolden (this repository's golden generator), to be
// copied into the r900 package of an rtlamr checkout next to the UNMODIFIED
// r900.go.  The second stage of the r900 parser -- filter(), r900.go:82-150 --
// and its buffers are unexported, and Parse shows the outside world only the
// messages that survive Reed-Solomon checking; this file lets the generator
// record what filter() computed on EVERY Decode call.
//
// AmdFilterStep performs exactly the buffer handling Parse does in front of
// its packet loop (r900.go:160-172: allocate once, slide p.signal by
// BlockSize, append Decoder.Signal[SymbolLength:]) and then calls the
// reference's own filter().  It returns p.quantized (aliased, valid until the
// next call).  Nothing here restates the arithmetic.
package r900

import "github.com/bemasher/rtlamr/protocol"

// AmdFilterStep must be called once after every protocol.Decoder.Decode-equivalent
// step of the decoder the parser was registered with (Parse runs once per call).
func AmdFilterStep(pp protocol.Parser) []byte {
	p := pp.(*Parser)
	p.once.Do(func() {
		p.cfg = p.Decoder.Cfg
		p.signal = make([]float32, p.Decoder.Cfg.BufferLength)
		p.csum = make([]float32, p.Decoder.Cfg.BufferLength+1)
		p.quantized = make([]byte, p.Decoder.Cfg.BufferLength)
	})
	cfg := p.cfg
	copy(p.signal, p.signal[cfg.BlockSize:])
	copy(p.signal[cfg.PacketLength:], p.Decoder.Signal[cfg.SymbolLength:])
	p.filter()
	return p.quantized
}
