//go:build mpi_collectives_in_place

// (the build tag keeps `go vet ./...` of THIS module from compiling the file where it lies: it is a file of
// package mpi and only builds inside the reference's directory -- remove the tag line when copying it there)
//
// collectives.go -- the file a maintainer drops INTO the reference package (github.com/btracey/mpi,
// next to mpi.go) to give it the collective entry points it only stubs today
// (`//func AllReduce() {}`, mpi.go:130).  Same delegate style as mpi.go:96-159; the optional
// interface is probed on the registered backend, which is what the unused `isAllReducer`
// variable (mpi.go:69-71) was evidently reserved for.  NOT compiled here (no Go toolchain).
package mpi

import "errors"

// Collective is implemented by backends that provide collectives natively (xgmi.Backend does).
// Buffers are whatever the backend accepts as `data` in Send/Receive (xgmi.DeviceBuffer on the
// MI355X backend); op is the backend's reduction operator (xgmi.Sum, ...).
type Collective interface {
	Bcast(buf interface{}, root int) error
	Reduce(send, recv interface{}, op int, root int) error
	Allreduce(send, recv interface{}, op int) error
	Allgather(send, recv interface{}) error
	Barrier() error
}

var errNoCollectives = errors.New("mpi: the registered implementation provides no collectives")

// Bcast replicates root's buffer on every rank.
func Bcast(buf interface{}, root int) error {
	if c, ok := mpier.(Collective); ok {
		return c.Bcast(buf, root)
	}
	return errNoCollectives
}

// Reduce folds every rank's send buffer into recv on root.
func Reduce(send, recv interface{}, op int, root int) error {
	if c, ok := mpier.(Collective); ok {
		return c.Reduce(send, recv, op, root)
	}
	return errNoCollectives
}

// Allreduce folds every rank's send buffer into recv on every rank.
func Allreduce(send, recv interface{}, op int) error {
	if c, ok := mpier.(Collective); ok {
		return c.Allreduce(send, recv, op)
	}
	return errNoCollectives
}

// Allgather concatenates every rank's send buffer, in rank order, into recv on every rank.
func Allgather(send, recv interface{}) error {
	if c, ok := mpier.(Collective); ok {
		return c.Allgather(send, recv)
	}
	return errNoCollectives
}

// Barrier blocks until every rank has called it.
func Barrier() error {
	if c, ok := mpier.(Collective); ok {
		return c.Barrier()
	}
	return errNoCollectives
}
