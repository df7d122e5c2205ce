// Package xgmi is the cgo binding that puts the MI355X backend (libxmpi.so, include/xmpi.h) behind
// btracey/mpi's Go API.  It implements mpi.Interface (mpi.go:163-170), so a program switches
// backends with one line in package main and nothing else changes:
//
//	func init() { mpi.Register(&xgmi.Backend{}) }     // mpi.go:61-67
//
// plus the collectives the reference only stubs out (mpi.go:130).
//
// NOTE: the build image of this repository has no Go toolchain, so this file is shipped as
// source and has NOT been compiled here.  It is deliberately thin: one cgo call per method and
// no logic beyond argument marshalling; the same C entry points are exercised from C++
// (mpi_amd/host) and Python (mpi_amd/xmpi.py) by the test-suite.
package xgmi

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../mpi_amd -lxmpi -Wl,-rpath,${SRCDIR}/../../mpi_amd
#include <stdlib.h>
#include "xmpi.h"
*/
import "C"

import (
	"bytes"
	"encoding/gob"
	"errors"
	"fmt"
	"hash/fnv"
	"os"
	"sort"
	"strconv"
	"strings"
	"time"
	"unsafe"

	"github.com/btracey/mpi"
)

// Float16 is an IEEE binary16 bit pattern (Go has no float16 type).
type Float16 uint16

// DType mirrors xmpi_dtype.
type DType int

const (
	U8 DType = iota
	I32
	I64
	F16
	F32
	F64
	BF16
)

// Op mirrors xmpi_op.
type Op int

const (
	Sum Op = iota
	Prod
	Min
	Max
)

// DeviceBuffer is a typed span of this rank's HBM; it is what `data interface{}` carries on the
// hot path (the payload never leaves device memory).
type DeviceBuffer struct {
	Ptr   unsafe.Pointer
	Count int
	Type  DType
}

// Backend replaces mpi.Network (network.go:25-39).  Zero-valued fields are filled from the
// package flags of btracey/mpi (network.go:69-90).
type Backend struct {
	Addr     string
	Addrs    []string
	Timeout  time.Duration
	Password string
	Device   int // 0 (zero value): $XMPI_DEVICE, else rank % visible GPUs; k > 0: GPU k-1
	Algo     int // xmpi_algo, 0 = auto

	comm *C.xmpi_comm
}

func status(rc C.int, where string) error {
	if rc == 0 {
		return nil
	}
	if rc == C.XMPI_ERR_TAG_EXISTS {
		return errors.New(C.GoString(C.xmpi_last_error())) // text of mpi.TagExists.Error()
	}
	return fmt.Errorf("%s: %s; %s", where, C.GoString(C.xmpi_strerror(rc)), C.GoString(C.xmpi_last_error()))
}

// Init implements mpi.Interface (replaces (*Network).Init, network.go:53-65).
func (b *Backend) Init() error {
	if b.Password == "" {
		b.Password = mpi.FlagPassword
	}
	if b.Timeout == 0 {
		b.Timeout = time.Duration(mpi.FlagInitTimeout)
	}
	if b.Addr == "" {
		b.Addr = mpi.FlagAddr
	}
	if len(b.Addrs) == 0 {
		b.Addrs = append([]string(nil), mpi.FlagAllAddrs...)
	}
	if len(b.Addrs) == 0 {
		b.Addr, b.Addrs = ":5000", []string{":5000"}
	}
	sort.Strings(b.Addrs) // rank = index in the sorted list (network.go:94-109)
	for i := 0; i+1 < len(b.Addrs); i++ {
		if b.Addrs[i] == b.Addrs[i+1] {
			return fmt.Errorf("network addresses not unique. list is: %v", b.Addrs)
		}
	}
	rank := sort.SearchStrings(b.Addrs, b.Addr)
	if !(rank < len(b.Addrs) && b.Addrs[rank] == b.Addr) {
		return fmt.Errorf("mpi init: local ip address not in global list. Local address is: %v, global list is %v", b.Addr, b.Addrs)
	}
	h := fnv.New64a()
	h.Write([]byte(strings.Join(b.Addrs, ",") + ","))
	h.Write([]byte(b.Password))
	key := os.Getenv("XMPI_JOB") + strconv.FormatUint(h.Sum64(), 16)
	device := b.Device - 1 // zero value = automatic
	if b.Device == 0 {
		device = -1 // libxmpi: rank % visible GPUs
		if os.Getenv("XMPI_DEVICE") != "" {
			device, _ = strconv.Atoi(os.Getenv("XMPI_DEVICE"))
		}
	}
	if b.Timeout > 0 {
		// -mpi-inittimeout bounds Init only (network.go:223-234,307-312); Send / Receive block for as long as it takes
		os.Setenv("XMPI_INIT_TIMEOUT_S", strconv.Itoa(int(b.Timeout.Seconds()+0.999)))
	}
	ckey := C.CString(key)
	defer C.free(unsafe.Pointer(ckey))
	return status(C.xmpi_init(C.int(rank), C.int(len(b.Addrs)), C.int(device), ckey, &b.comm), "mpi init")
}

// Finalize implements mpi.Interface.
func (b *Backend) Finalize() {
	if b.comm != nil {
		C.xmpi_finalize(b.comm)
		b.comm = nil
	}
}

// Rank implements mpi.Interface: -1 before Init (mpi.go:110-111).
func (b *Backend) Rank() int { return int(C.xmpi_rank(b.comm)) }

// Size implements mpi.Interface: 0 before Init.
func (b *Backend) Size() int { return int(C.xmpi_size(b.comm)) }

// view returns (pointer, count, dtype) for the payload types that travel without encoding.
func view(data interface{}) (unsafe.Pointer, int, DType, bool) {
	switch v := data.(type) {
	case DeviceBuffer:
		return v.Ptr, v.Count, v.Type, true
	case []byte:
		if len(v) == 0 {
			return nil, 0, U8, true
		}
		return unsafe.Pointer(&v[0]), len(v), U8, true
	case mpi.Raw:
		if len(v) == 0 {
			return nil, 0, U8, true
		}
		return unsafe.Pointer(&v[0]), len(v), U8, true
	case []int32:
		if len(v) == 0 {
			return nil, 0, I32, true
		}
		return unsafe.Pointer(&v[0]), len(v), I32, true
	case []int64:
		if len(v) == 0 {
			return nil, 0, I64, true
		}
		return unsafe.Pointer(&v[0]), len(v), I64, true
	case []Float16:
		if len(v) == 0 {
			return nil, 0, F16, true
		}
		return unsafe.Pointer(&v[0]), len(v), F16, true
	case []float32:
		if len(v) == 0 {
			return nil, 0, F32, true
		}
		return unsafe.Pointer(&v[0]), len(v), F32, true
	case []float64:
		if len(v) == 0 {
			return nil, 0, F64, true
		}
		return unsafe.Pointer(&v[0]), len(v), F64, true
	}
	return nil, 0, U8, false
}

// Send implements mpi.Interface (replaces (*Network).Send, network.go:518-572).  Numeric slices
// and DeviceBuffers travel as they are; anything else (e.g. the strings of helloworld.go) is
// gob-encoded like the reference does and sent as bytes.
func (b *Backend) Send(data interface{}, destination, tag int) error {
	if p, n, dt, ok := view(data); ok {
		return status(C.xmpi_send(b.comm, p, C.size_t(n), C.xmpi_dtype(dt), C.int(destination), C.int(tag)), "mpi send")
	}
	var buf bytes.Buffer
	if err := gob.NewEncoder(&buf).Encode(data); err != nil {
		return err
	}
	raw := buf.Bytes()
	var p unsafe.Pointer
	if len(raw) > 0 {
		p = unsafe.Pointer(&raw[0])
	}
	return status(C.xmpi_send(b.comm, p, C.size_t(len(raw)), C.xmpi_dtype(U8), C.int(destination), C.int(tag)), "mpi send")
}

// SendNoWait and Wait are the pair sketched in the comment block of mpi.go:132-152: SendNoWait
// returns once the payload has left `data` (numeric slices and DeviceBuffers only), Wait blocks
// until `destination` confirmed the message with `tag` and frees the {destination, tag} pair.
func (b *Backend) SendNoWait(data interface{}, destination, tag int) error {
	p, n, dt, ok := view(data)
	if !ok {
		return errPayload
	}
	return status(C.xmpi_send_nowait(b.comm, p, C.size_t(n), C.xmpi_dtype(dt), C.int(destination), C.int(tag)), "mpi send")
}

func (b *Backend) Wait(destination, tag int) error {
	return status(C.xmpi_wait(b.comm, C.int(destination), C.int(tag)), "mpi wait")
}

func (b *Backend) probe(source, tag int) (int, error) {
	var n C.size_t
	var dt C.xmpi_dtype
	err := status(C.xmpi_probe(b.comm, C.int(source), C.int(tag), &n, &dt), "mpi receive")
	return int(n), err
}

func (b *Backend) recvInto(p unsafe.Pointer, n int, dt DType, source, tag int) error {
	var got C.size_t
	return status(C.xmpi_recv(b.comm, p, C.size_t(n), C.xmpi_dtype(dt), C.int(source), C.int(tag), &got), "mpi receive")
}

// Receive implements mpi.Interface (replaces (*Network).Receive, network.go:575-602).  A pointer
// to a slice is re-sliced / re-allocated to the incoming length, as gob's in-place decode does
// (bounce.go:89,94).
func (b *Backend) Receive(data interface{}, source, tag int) error {
	switch v := data.(type) {
	case DeviceBuffer:
		return b.recvInto(v.Ptr, v.Count, v.Type, source, tag)
	case *[]byte:
		n, err := b.probe(source, tag)
		if err != nil {
			return err
		}
		if cap(*v) < n {
			*v = make([]byte, n)
		}
		*v = (*v)[:n]
		return b.recvInto(ptrOrNil(n, func() unsafe.Pointer { return unsafe.Pointer(&(*v)[0]) }), n, U8, source, tag)
	case *[]float32:
		n, err := b.probe(source, tag)
		if err != nil {
			return err
		}
		if cap(*v) < n {
			*v = make([]float32, n)
		}
		*v = (*v)[:n]
		return b.recvInto(ptrOrNil(n, func() unsafe.Pointer { return unsafe.Pointer(&(*v)[0]) }), n, F32, source, tag)
	case *[]float64:
		n, err := b.probe(source, tag)
		if err != nil {
			return err
		}
		if cap(*v) < n {
			*v = make([]float64, n)
		}
		*v = (*v)[:n]
		return b.recvInto(ptrOrNil(n, func() unsafe.Pointer { return unsafe.Pointer(&(*v)[0]) }), n, F64, source, tag)
	case *[]int64:
		n, err := b.probe(source, tag)
		if err != nil {
			return err
		}
		if cap(*v) < n {
			*v = make([]int64, n)
		}
		*v = (*v)[:n]
		return b.recvInto(ptrOrNil(n, func() unsafe.Pointer { return unsafe.Pointer(&(*v)[0]) }), n, I64, source, tag)
	}
	// anything else arrives gob-encoded (see Send)
	n, err := b.probe(source, tag)
	if err != nil {
		return err
	}
	raw := make([]byte, n)
	if err := b.recvInto(ptrOrNil(n, func() unsafe.Pointer { return unsafe.Pointer(&raw[0]) }), n, U8, source, tag); err != nil {
		return err
	}
	return gob.NewDecoder(bytes.NewBuffer(raw)).Decode(data)
}

func ptrOrNil(n int, f func() unsafe.Pointer) unsafe.Pointer {
	if n == 0 {
		return nil
	}
	return f()
}

// ---- collectives (absent upstream: mpi.go:130) --------------------------------------------------

var errPayload = errors.New("xgmi: collectives take a DeviceBuffer or a numeric slice")

// Bcast replicates root's buffer on every rank (bit-exact).  The signatures below are the
// mpi.Collective interface of go/mpi_collectives/collectives.go.
func (b *Backend) Bcast(buf interface{}, root int) error {
	p, n, dt, ok := view(buf)
	if !ok {
		return errPayload
	}
	return status(C.xmpi_bcast(b.comm, p, C.size_t(n), C.xmpi_dtype(dt), C.int(root), 0), "mpi bcast")
}

// Reduce folds the ranks' send buffers into recv on root.
func (b *Backend) Reduce(send, recv interface{}, op int, root int) error {
	sp, n, dt, ok1 := view(send)
	rp, _, _, ok2 := view(recv)
	if !ok1 || !ok2 {
		return errPayload
	}
	return status(C.xmpi_reduce(b.comm, sp, rp, C.size_t(n), C.xmpi_dtype(dt), C.xmpi_op(op), C.int(root), 0), "mpi reduce")
}

// Allreduce folds the ranks' send buffers into recv on every rank.
func (b *Backend) Allreduce(send, recv interface{}, op int) error {
	sp, n, dt, ok1 := view(send)
	rp, _, _, ok2 := view(recv)
	if !ok1 || !ok2 {
		return errPayload
	}
	return status(C.xmpi_allreduce(b.comm, sp, rp, C.size_t(n), C.xmpi_dtype(dt), C.xmpi_op(op), C.int(b.Algo)), "mpi allreduce")
}

// Allgather concatenates the ranks' send buffers, in rank order, into recv on every rank.
func (b *Backend) Allgather(send, recv interface{}) error {
	sp, n, dt, ok1 := view(send)
	rp, _, _, ok2 := view(recv)
	if !ok1 || !ok2 {
		return errPayload
	}
	return status(C.xmpi_allgather(b.comm, sp, rp, C.size_t(n), C.xmpi_dtype(dt), 0), "mpi allgather")
}

// Barrier is a host-side rendezvous of all ranks.
func (b *Backend) Barrier() error { return status(C.xmpi_barrier(b.comm), "mpi barrier") }

// IAllreduce starts an allreduce on DeviceBuffers and returns at once; the error arrives on the
// channel when the operation has completed (non-blocking collectives run in issue order on the
// communicator's worker: issue them in the same order on every rank).  Device buffers only: C keeps
// the pointers after the call returns, which cgo forbids for Go memory.
func (b *Backend) IAllreduce(send, recv DeviceBuffer, op int) <-chan error {
	done := make(chan error, 1)
	var req *C.xmpi_request
	rc := C.xmpi_iallreduce(b.comm, send.Ptr, recv.Ptr, C.size_t(send.Count), C.xmpi_dtype(send.Type), C.xmpi_op(op),
		C.int(b.Algo), &req)
	if err := status(rc, "mpi iallreduce"); err != nil {
		done <- err
		return done
	}
	go func() { done <- status(C.xmpi_request_wait(req), "mpi iallreduce") }()
	return done
}

// Register makes device memory that did not come from Malloc (another allocator's) reachable by the
// zero-copy collectives and the direct point-to-point path; Deregister before freeing it.
func (b *Backend) Register(ptr unsafe.Pointer, bytes int) error {
	return status(C.xmpi_register(b.comm, ptr, C.size_t(bytes)), "mpi register")
}
func (b *Backend) Deregister(ptr unsafe.Pointer) error {
	return status(C.xmpi_deregister(b.comm, ptr), "mpi deregister")
}

// Malloc allocates `count` elements of HBM on this rank's GPU: a block of an arena every peer maps
// once, so collectives and Send / Receive on it move the bytes GPU to GPU in one pass.
func (b *Backend) Malloc(count int, dt DType) DeviceBuffer {
	bytes := C.size_t(count) * C.xmpi_dtype_size(C.xmpi_dtype(dt))
	return DeviceBuffer{Ptr: C.xmpi_malloc(b.comm, bytes), Count: count, Type: dt}
}

// Free releases a buffer obtained from Malloc.
func (b *Backend) Free(buf DeviceBuffer) { C.xmpi_free(b.comm, buf.Ptr) }

// Upload / Download copy between a Go slice and HBM (blocking).
func (b *Backend) Upload(dst DeviceBuffer, src unsafe.Pointer, bytes int) error {
	return status(C.xmpi_memcpy(b.comm, dst.Ptr, src, C.size_t(bytes)), "mpi memcpy")
}
func (b *Backend) Download(dst unsafe.Pointer, src DeviceBuffer, bytes int) error {
	return status(C.xmpi_memcpy(b.comm, dst, src.Ptr, C.size_t(bytes)), "mpi memcpy")
}
