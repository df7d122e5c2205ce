// Package xgmi is the cgo binding that puts the MI355X backend (libxmpi.so, include/xmpi.h) behind
// btracey/mpi's Go API.  It implements mpi.Interface (mpi.go:163-170), so a program switches
// backends with one line in package main and nothing else changes:
//
//	func init() { mpi.Register(&xgmi.Backend{}) }     // mpi.go:61-67
//
// plus the collectives the reference only stubs out (mpi.go:130).
//
// NOTE: the build image of this repository has no Go toolchain, so this file is shipped as
// source and has NOT been compiled here (acceptance when one exists: `go vet ./...` in go/, then
// examples/helloworld and examples/bounce of the reference with the one-line Register).  It is
// deliberately thin: one cgo call per method and no logic beyond argument marshalling.  The calling
// pattern it imposes on the C ABI -- every call from an OS thread that never selected a device,
// stack out-parameters, nil pointers for empty slices -- is exercised in C by tests/cgo_shape_check.c;
// the same entry points are driven from C++ (mpi_amd/host) and Python (mpi_amd/xmpi.py) by the test-suite.
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
	"reflect"
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
	Algo     int // xmpi_algo, 0 = auto (the library's tuned table); e.g. AlgoRingPush

	comm *C.xmpi_comm
}

// The schedules a Backend can be told to use by name (xmpi.h xmpi_algo).  The *Push forms are the stepped kernels with every
// payload byte STORED over its link instead of loaded (bit-identical results): which a node's links move faster is measured by
// Tune, which times both.
const (
	AlgoAuto     = int(C.XMPI_ALGO_AUTO)
	AlgoRing     = int(C.XMPI_ALGO_RING)
	AlgoRHD      = int(C.XMPI_ALGO_RHD)
	AlgoDirect   = int(C.XMPI_ALGO_DIRECT)
	AlgoTree     = int(C.XMPI_ALGO_TREE)
	AlgoZcopy    = int(C.XMPI_ALGO_ZCOPY)
	AlgoZpush    = int(C.XMPI_ALGO_ZPUSH)
	AlgoLL       = int(C.XMPI_ALGO_LL)
	AlgoRingPush = int(C.XMPI_ALGO_RING_PUSH)
	AlgoRHDPush  = int(C.XMPI_ALGO_RHD_PUSH)
	AlgoTreePush = int(C.XMPI_ALGO_TREE_PUSH)
)

func status(rc C.int, where string) error {
	if rc == 0 {
		return nil
	}
	return fmt.Errorf("%s: %s; %s", where, C.GoString(C.xmpi_strerror(rc)), C.GoString(C.xmpi_last_error()))
}

// statusTag is status for the calls that take a tag: a {peer, tag} pair already in use is reported with the
// reference's own error type (mpi.go:172-182; the reference declares it and then panics instead, network.go:469).
func statusTag(rc C.int, where string, tag int) error {
	if rc == C.XMPI_ERR_TAG_EXISTS {
		return mpi.TagExists{Tag: tag}
	}
	return status(rc, where)
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

// sliceTypes is THE table of Go types that travel as raw typed payloads: Send (view) and Receive both
// consult it, so whatever one side sends raw the other side can receive raw.
var sliceTypes = map[reflect.Type]DType{
	reflect.TypeOf([]byte(nil)):    U8,
	reflect.TypeOf(mpi.Raw(nil)):   U8,
	reflect.TypeOf([]int32(nil)):   I32,
	reflect.TypeOf([]int64(nil)):   I64,
	reflect.TypeOf([]Float16(nil)): F16,
	reflect.TypeOf([]float32(nil)): F32,
	reflect.TypeOf([]float64(nil)): F64,
}

// view returns (pointer, count, dtype) for the payload types that travel without encoding.
func view(data interface{}) (unsafe.Pointer, int, DType, bool) {
	if v, ok := data.(DeviceBuffer); ok {
		return v.Ptr, v.Count, v.Type, true
	}
	if data == nil {
		return nil, 0, U8, false
	}
	rv := reflect.ValueOf(data)
	dt, ok := sliceTypes[rv.Type()]
	if !ok {
		return nil, 0, U8, false
	}
	if rv.Len() == 0 {
		return nil, 0, dt, true // empty slice: nil pointer, count 0 (the C side accepts that)
	}
	return rv.UnsafePointer(), rv.Len(), dt, true
}

// Send implements mpi.Interface (replaces (*Network).Send, network.go:518-572).  Numeric slices
// and DeviceBuffers travel as they are; anything else (e.g. the strings of helloworld.go) is
// gob-encoded like the reference does and sent as bytes.
func (b *Backend) Send(data interface{}, destination, tag int) error {
	if p, n, dt, ok := view(data); ok {
		return statusTag(C.xmpi_send(b.comm, p, C.size_t(n), C.xmpi_dtype(dt), C.int(destination), C.int(tag)), "mpi send", tag)
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
	return statusTag(C.xmpi_send(b.comm, p, C.size_t(len(raw)), C.xmpi_dtype(U8), C.int(destination), C.int(tag)), "mpi send", tag)
}

// SendNoWait and Wait are the pair sketched in the comment block of mpi.go:132-152: SendNoWait
// returns once the payload has left `data` (numeric slices and DeviceBuffers only), Wait blocks
// until `destination` confirmed the message with `tag` and frees the {destination, tag} pair.
func (b *Backend) SendNoWait(data interface{}, destination, tag int) error {
	p, n, dt, ok := view(data)
	if !ok {
		return errPayload
	}
	return statusTag(C.xmpi_send_nowait(b.comm, p, C.size_t(n), C.xmpi_dtype(dt), C.int(destination), C.int(tag)), "mpi send", tag)
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
	return statusTag(C.xmpi_recv(b.comm, p, C.size_t(n), C.xmpi_dtype(dt), C.int(source), C.int(tag), &got), "mpi receive", tag)
}

// Receive implements mpi.Interface (replaces (*Network).Receive, network.go:575-602).  A pointer
// to a slice is re-sliced / re-allocated to the incoming length, as gob's in-place decode does
// (bounce.go:89,94).
func (b *Backend) Receive(data interface{}, source, tag int) error {
	if v, ok := data.(DeviceBuffer); ok {
		return b.recvInto(v.Ptr, v.Count, v.Type, source, tag)
	}
	// *[]T for every T of sliceTypes (incl. *mpi.Raw): size the slice to the message, receive in place
	if rv := reflect.ValueOf(data); rv.Kind() == reflect.Ptr && !rv.IsNil() {
		if dt, ok := sliceTypes[rv.Elem().Type()]; ok {
			n, err := b.probe(source, tag)
			if err != nil {
				return err
			}
			s := rv.Elem()
			if s.Cap() < n {
				s.Set(reflect.MakeSlice(s.Type(), n, n))
			} else {
				s.SetLen(n)
			}
			var p unsafe.Pointer
			if n > 0 {
				p = s.UnsafePointer()
			}
			return b.recvInto(p, n, dt, source, tag)
		}
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

// Degraded reports what Init's vote left the job with: "" when every rank mapped every peer's memory, otherwise which level the job
// runs at and the first reason a rank gave (the reference's Init returns an error only when the mesh cannot be built,
// network.go:53-65 -- so does this one; a refused mapping degrades the job instead).
func (b *Backend) Degraded() string { return C.GoString(C.xmpi_degraded(b.comm)) }

// Tune lets the library time its own schedules on this job's GPUs and links for messages up to maxBytes per rank and keep
// the winner per size class: Allreduce / Allgather (and the stream-ordered forms) follow that table from then on.  Collective:
// every rank calls it with the same maxBytes, once, after Init.  (XMPI_AUTOTUNE_BYTES=N in the environment does the same
// inside Init for a program that is not to be touched.)
func (b *Backend) Tune(maxBytes int) error {
	return status(C.xmpi_tune(b.comm, C.size_t(maxBytes)), "mpi tune")
}

// SendOnStream / ReceiveOnStream are Send and Receive as ONE kernel each, enqueued on a HIP stream (the message and the ack
// of network.go:562-571, 616-624 as two 64-byte records in HBM, the payload pulled straight out of the sender's buffer):
// no host thread waits; StreamSync reports what went wrong.  Device buffers only.  A kernel that waits holds its stream:
// enqueue matching sends and receives so that no stream waits for work queued behind it.
func (b *Backend) SendOnStream(buf DeviceBuffer, destination, tag int, stream unsafe.Pointer) error {
	return status(C.xmpi_send_on_stream(b.comm, buf.Ptr, C.size_t(buf.Count), C.xmpi_dtype(buf.Type), C.int(destination), C.int(tag), stream), "mpi send")
}
func (b *Backend) ReceiveOnStream(buf DeviceBuffer, source, tag int, stream unsafe.Pointer) error {
	return status(C.xmpi_recv_on_stream(b.comm, buf.Ptr, C.size_t(buf.Count), C.xmpi_dtype(buf.Type), C.int(source), C.int(tag), stream), "mpi receive")
}

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

// Stream-ordered collectives (xmpi_*_on_stream): enqueued on a HIP stream of this rank's GPU like a kernel
// launch -- the call returns without waiting for any peer, one kernel per rank is the whole collective.  Device
// buffers only.  Stream() creates a stream (nil = the communicator's own); StreamSync waits for it and returns
// the status of the collectives that ran on it.  The overlap sketched at mpi.go:132-152, on streams.
func (b *Backend) Stream() unsafe.Pointer { return C.xmpi_stream_create(b.comm) }
func (b *Backend) StreamDestroy(stream unsafe.Pointer) {
	C.xmpi_stream_destroy(b.comm, stream)
}
func (b *Backend) StreamSync(stream unsafe.Pointer) error {
	return status(C.xmpi_stream_sync(b.comm, stream), "mpi stream sync")
}
func (b *Backend) AllreduceOnStream(send, recv DeviceBuffer, op int, stream unsafe.Pointer) error {
	return status(C.xmpi_allreduce_on_stream(b.comm, send.Ptr, recv.Ptr, C.size_t(send.Count), C.xmpi_dtype(send.Type),
		C.xmpi_op(op), stream), "mpi allreduce")
}
func (b *Backend) AllgatherOnStream(send, recv DeviceBuffer, stream unsafe.Pointer) error {
	return status(C.xmpi_allgather_on_stream(b.comm, send.Ptr, recv.Ptr, C.size_t(send.Count), C.xmpi_dtype(send.Type),
		stream), "mpi allgather")
}
func (b *Backend) BcastOnStream(buf DeviceBuffer, root int, stream unsafe.Pointer) error {
	return status(C.xmpi_bcast_on_stream(b.comm, buf.Ptr, C.size_t(buf.Count), C.xmpi_dtype(buf.Type), C.int(root), stream), "mpi bcast")
}
func (b *Backend) ReduceOnStream(send, recv DeviceBuffer, op int, root int, stream unsafe.Pointer) error {
	return status(C.xmpi_reduce_on_stream(b.comm, send.Ptr, recv.Ptr, C.size_t(send.Count), C.xmpi_dtype(send.Type),
		C.xmpi_op(op), C.int(root), stream), "mpi reduce")
}

// GraphBegin / GraphEnd capture the stream-ordered collectives enqueued on `stream` in between (and anything else the
// caller enqueues there) into an executable hipGraph; GraphLaunch replays it.  Every rank captures the same sequence
// and replays it equally often.
func (b *Backend) GraphBegin(stream unsafe.Pointer) error {
	return status(C.xmpi_graph_begin(b.comm, stream), "mpi graph begin")
}
func (b *Backend) GraphEnd(stream unsafe.Pointer) (unsafe.Pointer, error) {
	var g unsafe.Pointer
	err := status(C.xmpi_graph_end(b.comm, stream, &g), "mpi graph end")
	return g, err
}
func (b *Backend) GraphLaunch(graph, stream unsafe.Pointer) error {
	return status(C.xmpi_graph_launch(b.comm, graph, stream), "mpi graph launch")
}
func (b *Backend) GraphDestroy(graph unsafe.Pointer) {
	C.xmpi_graph_destroy(b.comm, graph)
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
