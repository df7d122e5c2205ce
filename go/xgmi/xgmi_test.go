// Tests of the cgo backend on a machine with a Go toolchain and an MI355X (NOT run in this repository's image, which
// has neither Go nor -- where the tests run -- a network; INTEGRATION.md section 3):
//
//	cd go && LD_LIBRARY_PATH=../mpi_amd go test ./xgmi
//
// One process, one rank: Send to oneself with a concurrent Receive is the reference's own local path
// (network.go:388-446, helloworld.go:53-81 uses it), so every entry of sliceTypes can make the round trip without a
// launcher.  With two processes under the launcher (`xmpirun 2 go test ./xgmi -run Pair`) the Pair tests run too.
package xgmi

import (
	"reflect"
	"sync"
	"testing"

	"github.com/btracey/mpi"
)

func backend(t *testing.T) *Backend {
	t.Helper()
	b := &Backend{}
	if err := b.Init(); err != nil {
		t.Skipf("no usable HIP device / job: %v", err)
	}
	return b
}

// every type of the table travels raw, arrives with its length, and is bit-identical
func TestSliceTypesRoundTrip(t *testing.T) {
	b := backend(t)
	defer b.Finalize()
	me := b.Rank()
	samples := []interface{}{
		[]byte{0, 1, 2, 250, 255}, mpi.Raw{9, 8, 7}, []int32{-1, 0, 1 << 30}, []int64{-1 << 62, 0, 1<<62 + 5},
		[]Float16{0x3c00, 0x7bff, 0x8001}, []float32{-0.0, 1.5, 3.4028235e38}, []float64{5e-324, -17, 1.7976931348623157e308},
		[]byte{}, []float64{},
	}
	if len(samples) < len(sliceTypes) {
		t.Fatalf("a type of sliceTypes has no sample")
	}
	for tag, s := range samples {
		if _, ok := sliceTypes[reflect.TypeOf(s)]; !ok {
			t.Fatalf("%T is not in sliceTypes", s)
		}
		got := reflect.New(reflect.TypeOf(s)) // *[]T, nil slice: Receive must size it
		var wg sync.WaitGroup
		var sendErr, recvErr error
		wg.Add(2)
		go func() { defer wg.Done(); sendErr = b.Send(s, me, tag) }()
		go func() { defer wg.Done(); recvErr = b.Receive(got.Interface(), me, tag) }()
		wg.Wait()
		if sendErr != nil || recvErr != nil {
			t.Fatalf("%T: send %v, receive %v", s, sendErr, recvErr)
		}
		if reflect.ValueOf(s).Len() != got.Elem().Len() || (got.Elem().Len() > 0 && !reflect.DeepEqual(s, got.Elem().Interface())) {
			t.Errorf("%T: sent %v, received %v", s, s, got.Elem().Interface())
		}
	}
	// anything else is gob-encoded like the reference does (helloworld's strings)
	var back string
	var wg sync.WaitGroup
	wg.Add(2)
	go func() { defer wg.Done(); _ = b.Send("\"I'm just node 0 talking to myself\"", me, 100) }()
	go func() { defer wg.Done(); _ = b.Receive(&back, me, 100) }()
	wg.Wait()
	if back != "\"I'm just node 0 talking to myself\"" {
		t.Errorf("string round trip: %q", back)
	}
}

// a second Send with a {destination, tag} that is still in flight is mpi.TagExists (mpi.go:172-182), not a panic
func TestTagExists(t *testing.T) {
	b := backend(t)
	defer b.Finalize()
	me := b.Rank()
	first := make(chan error, 1)
	go func() { first <- b.Send([]int64{1, 2, 3}, me, 7) }() // blocks: nobody receives yet
	for i := 0; i < 1000; i++ {                              // until the first send has registered its tag
		err := b.Send([]int64{4}, me, 7)
		if te, ok := err.(mpi.TagExists); ok {
			if te.Tag != 7 {
				t.Fatalf("TagExists.Tag = %d", te.Tag)
			}
			var got []int64
			if err := b.Receive(&got, me, 7); err != nil || len(got) != 3 {
				t.Fatalf("receive after the clash: %v %v", got, err)
			}
			if err := <-first; err != nil {
				t.Fatal(err)
			}
			return
		}
		if err == nil { // the second send got in first: drain both and try again
			t.Skip("the race went the other way")
		}
	}
	t.Fatal("no TagExists")
}

// Rank / Size before Init are the reference's "not initialised" answers (mpi.go:110-119)
func TestUninitialised(t *testing.T) {
	b := &Backend{}
	if b.Rank() != -1 || b.Size() != 0 {
		t.Fatalf("rank %d size %d before Init", b.Rank(), b.Size())
	}
}

// xmpirun 2 go test ./xgmi -run Pair: bounce.go's echo between two processes, every type of the table; then the
// collectives of go/mpi_collectives in rank order
func TestPairBounce(t *testing.T) {
	b := backend(t)
	defer b.Finalize()
	if b.Size() != 2 {
		t.Skip("needs two ranks (run under the launcher)")
	}
	peer := 1 - b.Rank()
	f := make([]float64, 100000)
	for i := range f {
		f[i] = float64(i) * 0.25
	}
	if b.Rank() == 0 {
		var back []float64
		if err := b.Send(f, peer, 0); err != nil {
			t.Fatal(err)
		}
		if err := b.Receive(&back, peer, 0); err != nil {
			t.Fatal(err)
		}
		if !reflect.DeepEqual(f, back) { // bounce.go:133 floats.Equal
			t.Error("echo differs")
		}
	} else {
		var got []float64
		if err := b.Receive(&got, peer, 0); err != nil {
			t.Fatal(err)
		}
		if err := b.Send(got, peer, 0); err != nil {
			t.Fatal(err)
		}
	}
	x := []float32{float32(b.Rank()) + 1, 0.5}
	sum := make([]float32, 2)
	if err := b.Allreduce(x, sum, int(Sum)); err != nil {
		t.Fatal(err)
	}
	if sum[0] != 3 || sum[1] != 1 {
		t.Errorf("allreduce: %v", sum)
	}
}
