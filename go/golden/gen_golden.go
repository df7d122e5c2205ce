// gen_golden runs the REFERENCE -- encoding/gob as network.go uses it, and mpi.Network itself over localhost TCP -- and
// writes what it produces as fixtures for this repository's tests (tests/test_reference_golden.py):
//
//	tests/golden/ref_gob.json          gob byte streams of the values the reference's programs send, of the handshake
//	                                   struct (network.go:198-201) and of the message / ack frames (network.go:511-514,
//	                                   562, 616-621), encoded exactly the way Send does it (network.go:539, 562)
//	tests/golden/ref_transcripts.json  what helloworld.go:53-81 and bounce.go:83-137 receive, and a rank-order allreduce a
//	                                   reference user composes from Send / Receive (the definition of this project's oracle),
//	                                   run through two or four mpi.Network instances talking TCP on localhost
//
// It pins the three things nothing in this repository's image could pin (no Go toolchain there): the codec restatements
// (oracle/gob_codec.h, mpi_amd/host/gobwire.hpp) against real gob bytes, the float arithmetic of the oracle's rank-order
// fold against Go's (every float32 operation rounded to float32), and the reference's Send / Receive end to end.
//
//	cd go && go run ./golden -out ../tests/golden && cd .. && python -m pytest tests/test_reference_golden.py
//
// NOT run in this repository's image (INTEGRATION.md section 3).  The input generator below restates oracle_fill
// (oracle/xmpi_oracle.c) so that the fixtures are the oracle's own test inputs.
package main

import (
	"bytes"
	"encoding/binary"
	"encoding/gob"
	"encoding/hex"
	"encoding/json"
	"flag"
	"fmt"
	"math"
	"os"
	"path/filepath"
	"sync"
	"time"

	"github.com/btracey/mpi"
)

// the reference's wire structs are unexported (network.go:198-201, 511-514): gob identifies a struct by its NAME and its
// fields, not by its package, so these twins encode to the same bytes
type initialMessage struct {
	Password string
	Id       int
}

type message struct {
	Tag   int
	Bytes mpi.Raw
}

func enc(v interface{}) string {
	var buf bytes.Buffer
	if err := gob.NewEncoder(&buf).Encode(v); err != nil {
		panic(err)
	}
	return hex.EncodeToString(buf.Bytes())
}

func le(v interface{}) string {
	var buf bytes.Buffer
	if err := binary.Write(&buf, binary.LittleEndian, v); err != nil {
		panic(err)
	}
	return hex.EncodeToString(buf.Bytes())
}

// oracle_hash / oracle_fill pattern 0 ("uniform") of oracle/xmpi_oracle.c
func hash64(seed, i uint64) uint64 {
	z := seed*0xD1342543DE82EF95 + i*0x9E3779B97F4A7C15 + 0x2545F4914F6CDD1D
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
	z = (z ^ (z >> 27)) * 0x94D049BB133111EB
	return z ^ (z >> 31)
}

func fillF32(n int, seed uint64) []float32 {
	out := make([]float32, n)
	for i := range out {
		out[i] = float32(float64(hash64(seed, uint64(i))>>40) * (1.0 / (1 << 24)))
	}
	return out
}

func fillF64(n int, seed uint64) []float64 {
	out := make([]float64, n)
	for i := range out {
		out[i] = float64(hash64(seed, uint64(i))>>11) * (1.0 / (1 << 53))
	}
	return out
}

func fillI64(n int, seed uint64) []int64 {
	out := make([]int64, n)
	for i := range out {
		out[i] = int64(hash64(seed, uint64(i)))
	}
	return out
}

// signed, cancelling values: pattern 3 of oracle_fill
func fillSignedF32(n int, seed uint64) []float32 {
	out := make([]float32, n)
	for i := range out {
		h := hash64(seed, uint64(i))
		m := float64((h>>40)&0xFF)*(1.0/256) - 0.5
		sh := int((h>>8)&7) - 4
		out[i] = float32(math.Ldexp(m, sh))
	}
	return out
}

type gobCase struct {
	Name   string `json:"name"`
	GoType string `json:"go_type"`
	Raw    string `json:"raw_le_hex"` // the values, little-endian (strings / []byte: the bytes)
	Gob    string `json:"gob_hex"`    // gob.NewEncoder(&buf).Encode(value): what Send puts into message.Bytes
}

func gobCases() []gobCase {
	var out []gobCase
	add := func(name, typ, raw string, v interface{}) { out = append(out, gobCase{name, typ, raw, enc(v)}) }
	for _, n := range []int{0, 1, 10, 1000} { // bounce.go:33's lengths, shortened
		b := make([]byte, n)
		for i := range b {
			b[i] = byte(hash64(1, uint64(i)) >> 56)
		}
		add(fmt.Sprintf("bytes_%d", n), "[]byte", hex.EncodeToString(b), b)
		f := fillF64(n, 2)
		add(fmt.Sprintf("float64_%d", n), "[]float64", le(f), f)
		g := fillF32(n, 3)
		add(fmt.Sprintf("float32_%d", n), "[]float32", le(g), g)
		q := fillI64(n, 4)
		add(fmt.Sprintf("int64_%d", n), "[]int64", le(q), q)
		w := make([]int32, n)
		for i := range w {
			w[i] = int32(uint32(hash64(5, uint64(i)) >> 32))
		}
		add(fmt.Sprintf("int32_%d", n), "[]int32", le(w), w)
	}
	edge64 := []float64{0, math.Copysign(0, -1), math.Inf(1), math.Inf(-1), math.SmallestNonzeroFloat64, math.MaxFloat64, 17, -129.5}
	add("float64_edges", "[]float64", le(edge64), edge64)
	edgeI := []int64{0, -1, 1, math.MaxInt64, math.MinInt64, 255, 256, -129}
	add("int64_edges", "[]int64", le(edgeI), edgeI)
	for _, s := range []string{"", "\"Hello node 1, I'm node 0\"", "\"I'm just node 0 talking to myself\""} { // helloworld.go:59-62
		add(fmt.Sprintf("string_%d", len(s)), "string", hex.EncodeToString([]byte(s)), s)
	}
	return out
}

type frameCase struct {
	Name     string `json:"name"`
	Tag      int    `json:"tag"`
	Id       int    `json:"id"`
	Password string `json:"password"`
	Payload  string `json:"payload_hex"`
	Gob      string `json:"gob_hex"`
}

func frameCases() []frameCase {
	var out []frameCase
	for _, id := range []int{0, 1, 7} {
		for _, pw := range []string{"", "secret"} {
			out = append(out, frameCase{Name: "initialMessage", Id: id, Password: pw, Gob: enc(initialMessage{Password: pw, Id: id})})
		}
	}
	payload := []byte{1, 2, 3, 250, 251, 252}
	for _, tag := range []int{0, 1, -5, 123456789} {
		out = append(out, frameCase{Name: "message", Tag: tag, Payload: hex.EncodeToString(payload), Gob: enc(message{Tag: tag, Bytes: payload})})
		out = append(out, frameCase{Name: "ack", Tag: tag, Gob: enc(message{Tag: tag})}) // network.go:616-621
	}
	return out
}

// n mpi.Network instances in this process, one goroutine each, talking TCP on localhost
func network(n, base int, body func(net *mpi.Network, rank int)) {
	addrs := make([]string, n)
	for i := range addrs {
		addrs[i] = fmt.Sprintf(":%d", base+i)
	}
	var wg sync.WaitGroup
	for i := 0; i < n; i++ {
		wg.Add(1)
		go func(i int) {
			defer wg.Done()
			net := &mpi.Network{Addr: addrs[i], Addrs: append([]string(nil), addrs...), Timeout: 20 * time.Second}
			if err := net.Init(); err != nil {
				panic(err)
			}
			body(net, net.Rank())
			net.Finalize()
		}(i)
	}
	wg.Wait()
}

type transcripts struct {
	Helloworld map[string][]string `json:"helloworld_received"` // rank -> messages in source order (helloworld.go:53-81)
	Bounce     []map[string]string `json:"bounce_echo"`         // per length: what came back, must equal what was sent (bounce.go:105,133)
	Allreduce  []map[string]string `json:"allreduce_rank_order"`
}

func runTranscripts() transcripts {
	var t transcripts
	var mu sync.Mutex
	t.Helloworld = map[string][]string{}
	network(3, 15000, func(net *mpi.Network, rank int) {
		size := net.Size()
		got := make([]string, size)
		var wg sync.WaitGroup
		for i := 0; i < size; i++ {
			wg.Add(2)
			go func(i int) {
				defer wg.Done()
				msg := fmt.Sprintf("\"Hello node %d, I'm node %d\"", i, rank)
				if i == rank {
					msg = fmt.Sprintf("\"I'm just node %d talking to myself\"", rank)
				}
				if err := net.Send(msg, i, 0); err != nil {
					panic(err)
				}
			}(i)
			go func(i int) {
				defer wg.Done()
				if err := net.Receive(&got[i], i, 0); err != nil {
					panic(err)
				}
			}(i)
		}
		wg.Wait()
		mu.Lock()
		t.Helloworld[fmt.Sprint(rank)] = got
		mu.Unlock()
	})
	for _, n := range []int{0, 1, 10, 1000, 100000} {
		row := map[string]string{"length": fmt.Sprint(n)}
		network(2, 15100+n%7*10, func(net *mpi.Network, rank int) {
			b := make([]byte, n)
			for i := range b {
				b[i] = byte(hash64(11, uint64(i)) >> 56)
			}
			f := fillF64(n/8, 12)
			if rank == 0 {
				back, fback := make([]byte, 0), make([]float64, 0)
				if err := net.Send(b, 1, 0); err != nil {
					panic(err)
				}
				if err := net.Receive(&back, 1, 0); err != nil {
					panic(err)
				}
				if err := net.Send(f, 1, 1); err != nil {
					panic(err)
				}
				if err := net.Receive(&fback, 1, 1); err != nil {
					panic(err)
				}
				row["sent_bytes_hex"], row["echo_bytes_hex"] = hex.EncodeToString(b), hex.EncodeToString(back)
				row["sent_f64_le_hex"], row["echo_f64_le_hex"] = le(f), le(fback)
			} else {
				var back []byte
				var fback []float64
				if err := net.Receive(&back, 0, 0); err != nil {
					panic(err)
				}
				if err := net.Send(back, 0, 0); err != nil {
					panic(err)
				}
				if err := net.Receive(&fback, 0, 1); err != nil {
					panic(err)
				}
				if err := net.Send(fback, 0, 1); err != nil {
					panic(err)
				}
			}
		})
		t.Bounce = append(t.Bounce, row)
	}
	// The oracle's definition of an allreduce, executed by the reference: every rank sends its whole buffer to every rank
	// (the helloworld idiom) and folds what it received in RANK ORDER 0..N-1, in the element type (Go rounds every float32
	// operation to float32).  Inputs: oracle_fill(pattern 0 / 3, seed 1000 + rank).
	for _, c := range []struct {
		name  string
		ranks int
		count int
	}{{"f32_uniform", 4, 1000}, {"f32_signed", 4, 1000}, {"f64_uniform", 3, 500}, {"i64_uniform", 4, 300}} {
		row := map[string]string{"name": c.name, "ranks": fmt.Sprint(c.ranks), "count": fmt.Sprint(c.count), "seed0": "1000"}
		network(c.ranks, 15300, func(net *mpi.Network, rank int) {
			size := net.Size()
			var mine interface{}
			switch c.name {
			case "f32_uniform":
				mine = fillF32(c.count, uint64(1000+rank))
			case "f32_signed":
				mine = fillSignedF32(c.count, uint64(1000+rank))
			case "f64_uniform":
				mine = fillF64(c.count, uint64(1000+rank))
			default:
				mine = fillI64(c.count, uint64(1000+rank))
			}
			got32, got64, gotI := make([][]float32, size), make([][]float64, size), make([][]int64, size)
			var wg sync.WaitGroup
			for i := 0; i < size; i++ {
				wg.Add(2)
				go func(i int) {
					defer wg.Done()
					if err := net.Send(mine, i, 0); err != nil {
						panic(err)
					}
				}(i)
				go func(i int) {
					defer wg.Done()
					var err error
					switch mine.(type) {
					case []float32:
						err = net.Receive(&got32[i], i, 0)
					case []float64:
						err = net.Receive(&got64[i], i, 0)
					default:
						err = net.Receive(&gotI[i], i, 0)
					}
					if err != nil {
						panic(err)
					}
				}(i)
			}
			wg.Wait()
			var result string
			switch mine.(type) {
			case []float32:
				acc := append([]float32(nil), got32[0]...)
				for r := 1; r < size; r++ {
					for i := range acc {
						acc[i] = acc[i] + got32[r][i]
					}
				}
				result = le(acc)
			case []float64:
				acc := append([]float64(nil), got64[0]...)
				for r := 1; r < size; r++ {
					for i := range acc {
						acc[i] = acc[i] + got64[r][i]
					}
				}
				result = le(acc)
			default:
				acc := append([]int64(nil), gotI[0]...)
				for r := 1; r < size; r++ {
					for i := range acc {
						acc[i] = acc[i] + gotI[r][i]
					}
				}
				result = le(acc)
			}
			if rank == 0 {
				row["result_le_hex"] = result
				row["input_rank0_le_hex"] = le(mine)
			}
		})
		t.Allreduce = append(t.Allreduce, row)
	}
	return t
}

func write(dir, name string, v interface{}) {
	b, err := json.MarshalIndent(v, "", " ")
	if err != nil {
		panic(err)
	}
	if err := os.WriteFile(filepath.Join(dir, name), append(b, '\n'), 0o644); err != nil {
		panic(err)
	}
	fmt.Println("wrote", filepath.Join(dir, name))
}

func main() {
	out := flag.String("out", "../tests/golden", "directory of the fixtures")
	flag.Parse()
	write(*out, "ref_gob.json", map[string]interface{}{
		"generator": "go/golden/gen_golden.go: encoding/gob of this toolchain; the reference's wire structs restated by name",
		"values":    gobCases(), "frames": frameCases()})
	write(*out, "ref_transcripts.json", runTranscripts())
}
