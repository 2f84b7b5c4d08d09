//go:build amdgolden

package main

import (
	"crypto/sha256"
	"encoding/hex"
	"encoding/json"
	"fmt"
	"log"
	"os"
	"path/filepath"

	"github.com/bemasher/rtlamr/r900"
)

// r900Case is one entry of tests/golden/r900_filter.json (made by the oracle) /
// go_r900_filter.json (made here by the reference's own r900.Parser.filter).
type r900Case struct {
	Name      string   `json:"name"`
	Protocols []string `json:"protocols"`
	Chip      int      `json:"chip"`
	Input     string   `json:"input"` // "capture" = assets/sample.bin, "synth" = the stream of the same name in synth.json
	Calls     int      `json:"calls"`
	QSha      string   `json:"qsha"` // sha256 over p.quantized after every Decode call, all calls concatenated
	Hist      [6]int64 `json:"hist"` // how often each of the six symbol values occurred (the zero tail counts as 0)
}

// r900Run drives the reference decoder block by block like decodeStream and, after every
// call, lets the r900 parser do what Parse does in front of its packet loop
// (r900.AmdFilterStep: slide, append Decoder.Signal, filter()).
func r900Run(protos []string, chip int, iq []byte) (string, [6]int64, int) {
	r := newRefDecoder(protos, chip)
	p, ok := r.parsers["r900"]
	if !ok {
		log.Fatal("r900Run: r900 is not among the protocols")
	}
	bs2 := r.d.Cfg.BlockSize2
	h := sha256.New()
	var hist [6]int64
	var q []byte
	var hits []hit
	n := 0
	for k := 0; (k+1)*bs2 <= len(iq); k++ {
		q, hits = q[:0], hits[:0]
		r.step(k, iq[k*bs2:(k+1)*bs2], &q, &hits)
		sym := r900.AmdFilterStep(p)
		h.Write(sym)
		for _, v := range sym {
			hist[v]++
		}
		n++
	}
	return hex.EncodeToString(h.Sum(nil)), hist, n
}

func r900Golden(golden string, capture []byte) {
	var have struct {
		Cases []r900Case `json:"cases"`
	}
	raw, err := os.ReadFile(filepath.Join(golden, "r900_filter.json"))
	if err != nil {
		log.Fatal(err)
	}
	if err := json.Unmarshal(raw, &have); err != nil {
		log.Fatal(err)
	}
	var synth struct {
		Cases []synthCase `json:"cases"`
	}
	raw, err = os.ReadFile(filepath.Join(golden, "synth.json"))
	if err != nil {
		log.Fatal(err)
	}
	if err := json.Unmarshal(raw, &synth); err != nil {
		log.Fatal(err)
	}
	var out struct {
		Generator string     `json:"generator"`
		Cases     []r900Case `json:"cases"`
	}
	out.Generator = "go/cmd/amdgolden -tags amdgolden over github.com/bemasher/rtlamr/r900 (reference Parser.filter)"
	for _, c := range have.Cases {
		iq := capture
		if c.Input == "synth" {
			iq = nil
			for _, s := range synth.Cases {
				if s.Name == c.Name {
					probe := newRefDecoder(s.Protocols, s.Chip)
					iq = synthStream(s.Protocols, s.Chip, s.Blocks, probe.d.Cfg.BlockSize, s.Seed, s.Packets)
					if got := shaHex(iq); got != s.IqSha {
						log.Fatalf("%s: synthetic stream differs from the recorded one", c.Name)
					}
				}
			}
			if iq == nil {
				log.Fatalf("%s: no such case in synth.json", c.Name)
			}
		}
		g := c
		g.QSha, g.Hist, g.Calls = r900Run(c.Protocols, c.Chip, iq)
		out.Cases = append(out.Cases, g)
		fmt.Printf("%-18s r900 filter over %4d calls  %s\n", c.Name, g.Calls, g.QSha[:16])
	}
	writeJSON(filepath.Join(golden, "go_r900_filter.json"), out)
}
