// Module of the cgo binding (package xgmi) that puts libxmpi.so behind github.com/btracey/mpi.
//
// The reference predates Go modules (no go.mod upstream, GOPATH-era imports), so the requirement below is met
// with a local checkout: give the reference a module file once
//     cd /path/to/btracey-mpi && go mod init github.com/btracey/mpi
// (its one third-party import, github.com/gonum/floats, is used by examples/bounce only), drop
// mpi_collectives/collectives.go of this directory next to its mpi.go, and point the replace line at it.
// Build libxmpi.so first (python -m mpi_amd.build); acceptance: `go vet ./...` here, then the reference's
// examples/helloworld and examples/bounce with `func init() { mpi.Register(&xgmi.Backend{}) }` added.
// NOT built in this repository's image: it has no Go toolchain (see INTEGRATION.md section 3).
module github.com/btracey/mpi-xgmi

go 1.18

require github.com/btracey/mpi v0.0.0

replace github.com/btracey/mpi => ../../reference
