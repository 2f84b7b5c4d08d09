//go:build amd

// Parity test of the MI355X decoder (decode_amd.go) against golden vectors the
// REFERENCE decoder produced (cmd/amdgolden).  The reference has no test for
// protocol.Decoder; this is the one it would need to keep a second
// implementation honest.
//
//	go run ./cmd/amdgolden -repo <repo> -iqdir /tmp/amriq      # reference build (no tag), writes the goldens
//	AMR_GOLDEN_DIR=<repo>/tests/golden AMR_IQ_DIR=/tmp/amriq \
//	    go test -tags amd ./protocol/ -run TestAmd               # MI355X build against them
package protocol_test

import (
	"crypto/sha256"
	"encoding/binary"
	"encoding/hex"
	"encoding/json"
	"os"
	"path/filepath"
	"sort"
	"strconv"
	"strings"
	"sync"
	"testing"

	"github.com/bemasher/rtlamr/protocol"

	_ "github.com/bemasher/rtlamr/idm"
	_ "github.com/bemasher/rtlamr/netidm"
	_ "github.com/bemasher/rtlamr/r900"
	_ "github.com/bemasher/rtlamr/scm"
	_ "github.com/bemasher/rtlamr/scmplus"
)

// recorder stands in for a parser: same Cfg, and Parse keeps what Decode hands it.
type recorder struct {
	protocol.Parser
	mu   *sync.Mutex
	pkts *[][]protocol.Data // one entry per Parse call = per Decode call
}

func (r recorder) Parse(pkts []protocol.Data, _ chan protocol.Message, wg *sync.WaitGroup) {
	r.mu.Lock()
	*r.pkts = append(*r.pkts, pkts)
	r.mu.Unlock()
	wg.Done()
}

type hit struct {
	pid, call, idx int
	bytes          []byte
}

type run struct {
	cfg  protocol.PacketConfig
	q    []byte
	hits []hit
}

func decodeAmd(t *testing.T, protos []string, chip int, iq []byte, blocksPerCall int, devices ...int) run {
	d := protocol.NewDecoder()
	d.KeepQuantized = len(devices) < 2 // (a diagnostic of one device: see Decoder.Devices)
	d.Devices = devices
	var mu sync.Mutex
	var recs []*[][]protocol.Data // per distinct preamble, in order of first registration
	seen := map[string]bool{}
	for _, name := range protos {
		p, err := protocol.NewParser(name, chip)
		if err != nil {
			t.Fatal(err)
		}
		store := new([][]protocol.Data)
		if !seen[p.Cfg().Preamble] { // the second parser of a preamble receives the same packets
			seen[p.Cfg().Preamble] = true
			recs = append(recs, store)
		}
		d.RegisterProtocol(recorder{p, &mu, store})
	}
	d.Allocate()
	defer d.Close()
	var out run
	out.cfg = d.Cfg
	bs, bs2, pl := d.Cfg.BlockSize, d.Cfg.BlockSize2, d.Cfg.PacketLength
	nBlocks := len(iq) / bs2
	for k := 0; k < nBlocks; k += blocksPerCall {
		n := blocksPerCall
		if k+n > nBlocks {
			n = nBlocks - k
		}
		for range d.Decode(iq[k*bs2 : (k+n)*bs2]) { // drain to the close, as main.go:235-262 does
		}
		if n == 1 { // Quantized holds this call's decisions behind the PacketLength history
			fresh := d.Quantized[pl : pl+bs]
			for i := 0; i < bs; i += 8 {
				var b byte
				for j := 0; j < 8; j++ {
					b = b<<1 | fresh[i+j]
				}
				out.q = append(out.q, b)
			}
		}
	}
	for pid, store := range recs {
		for call, pkts := range *store {
			for _, data := range pkts {
				out.hits = append(out.hits, hit{pid, call, data.Idx, data.Bytes})
			}
		}
	}
	sort.SliceStable(out.hits, func(a, b int) bool {
		x, y := out.hits[a], out.hits[b]
		if x.pid != y.pid {
			return x.pid < y.pid
		}
		if x.call != y.call {
			return x.call < y.call
		}
		return x.idx < y.idx
	})
	return out
}

func shaHex(b []byte) string {
	s := sha256.Sum256(b)
	return hex.EncodeToString(s[:])
}

func (r run) hitsSha() string {
	var buf []byte
	for _, h := range r.hits {
		for _, v := range []int{h.pid, h.call, h.idx} {
			buf = binary.LittleEndian.AppendUint64(buf, uint64(int64(v)))
		}
	}
	return shaHex(buf)
}

func (r run) pktSha(nbytes int) string {
	var buf []byte
	for _, h := range r.hits {
		buf = append(buf, h.bytes[:nbytes]...)
	}
	return shaHex(buf)
}

type synthCase struct {
	Name      string   `json:"name"`
	Protocols []string `json:"protocols"`
	Chip      int      `json:"chip"`
	QSha      string   `json:"qsha"`
	NHits     int      `json:"n_hits"`
	HitsSha   string   `json:"hits_sha"`
	PktSha    string   `json:"pkt_sha"`
}

func TestAmdMatchesReferenceGoldens(t *testing.T) {
	golden, iqdir := os.Getenv("AMR_GOLDEN_DIR"), os.Getenv("AMR_IQ_DIR")
	if golden == "" || iqdir == "" {
		t.Skip("AMR_GOLDEN_DIR / AMR_IQ_DIR not set (see the file comment)")
	}
	raw, err := os.ReadFile(filepath.Join(golden, "go_synth.json"))
	if err != nil {
		t.Fatal(err)
	}
	var g struct {
		Cases []synthCase `json:"cases"`
	}
	if err := json.Unmarshal(raw, &g); err != nil {
		t.Fatal(err)
	}
	for _, c := range g.Cases {
		iq, err := os.ReadFile(filepath.Join(iqdir, c.Name+".iq"))
		if err != nil {
			t.Fatal(err)
		}
		// one block per Decode call, exactly as main.go:235 drives the reference
		r := decodeAmd(t, c.Protocols, c.Chip, iq, 1)
		if got := shaHex(r.q); got != c.QSha {
			t.Errorf("%s: quantized bits differ from the reference (sha %s, want %s)", c.Name, got, c.QSha)
		}
		if len(r.hits) != c.NHits || r.hitsSha() != c.HitsSha {
			t.Errorf("%s: hit list differs from the reference (%d hits, want %d)", c.Name, len(r.hits), c.NHits)
		}
		if got := r.pktSha(r.cfg.PacketSymbols / 8); got != c.PktSha {
			t.Errorf("%s: packet bytes differ from the reference", c.Name)
		}
		// the same stream in batches of 37 blocks per call: identical hits (batching is invisible to the parsers)
		b := decodeAmd(t, c.Protocols, c.Chip, iq, 37)
		if len(b.hits) != c.NHits || b.hitsSha() != c.HitsSha || b.pktSha(b.cfg.PacketSymbols/8) != c.PktSha {
			t.Errorf("%s: batched decode differs from the reference", c.Name)
		}
	}
}

// TestDevicesMatchOneDevice: Decoder.Devices (one process, several GPUs: amr_comm_init_all, every Decode call's blocks cut
// into one primed range per device) against the same stream on one device -- every hit, every packet byte.  Needs
// AMR_TEST_DEVICES=0,1,... (at least two gfx950 ordinals); skipped otherwise.
func TestDevicesMatchOneDevice(t *testing.T) {
	var devs []int
	for _, f := range strings.Split(os.Getenv("AMR_TEST_DEVICES"), ",") {
		if v, err := strconv.Atoi(strings.TrimSpace(f)); err == nil {
			devs = append(devs, v)
		}
	}
	if len(devs) < 2 {
		t.Skip("AMR_TEST_DEVICES names fewer than two devices")
	}
	iq, err := os.ReadFile(filepath.Join("..", "assets", "sample.bin"))
	if err != nil {
		t.Skip("assets/sample.bin not found")
	}
	for _, per := range []int{64, 200, 1} { // ranges longer and shorter than the priming reach, and the main.go loop
		one := decodeAmd(t, []string{"scm", "idm"}, 72, iq, per)
		many := decodeAmd(t, []string{"scm", "idm"}, 72, iq, per, devs...)
		if len(one.hits) != len(many.hits) || one.hitsSha() != many.hitsSha() ||
			one.pktSha(one.cfg.PacketSymbols/8) != many.pktSha(many.cfg.PacketSymbols/8) {
			t.Errorf("%d blocks per call: %d devices give %d hits, one device %d (or other bytes)", per, len(devs), len(many.hits), len(one.hits))
		}
	}
}
