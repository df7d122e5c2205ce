#!/bin/bash
# pin_parity.sh -- ONE command for the day a Go toolchain exists: compiles the Go side of this repository against the reference,
# runs the REFERENCE to produce the golden fixtures (turning "parity unpinned" into pinned), and runs the reference's own two
# programs, source files untouched, on libxmpi.so through mpi.Register (mpi.go:56-67).
#
#   scripts/pin_parity.sh            REF=/path/to/btracey-mpi (default /root/reference; only READ: everything happens in a temp dir)
#
#   1. $T/ref   = a copy of $REF + `go mod init github.com/btracey/mpi` (the reference predates modules) + collectives.go (this
#                 repository's go/mpi_collectives/collectives.go, its build-tag line stripped: a file of package mpi, mpi.go:130)
#   2. $T/go    = a copy of go/ with `go mod edit -replace github.com/btracey/mpi=$T/ref`;  go vet ./... && go build ./...
#   3. go run ./golden -out tests/golden     the reference's gob streams and Send / Receive transcripts (go/golden/gen_golden.go)
#      python -m pytest tests/test_reference_golden.py     the oracle and the product codec against them, bit for bit
#   4. go test ./xgmi                        the cgo shim on the GPU (skipped with a message where no GPU is visible)
#   5. $T/run   = examples/helloworld/helloworld.go and examples/bounce/bounce.go of the reference, byte for byte, each beside a
#                 ONE-LINE file `func init() { mpi.Register(&xgmi.Backend{}) }`; built, and run under mpi_amd/bin/xmpirun
#                 (bounce needs github.com/gonum/floats: built when the module cache or the network has it, reported otherwise)
# Rehearsed in the CPU suite with a stub `go` on PATH that records its arguments (tests/test_host_mirror.py): the sequence below,
# and that $REF is never written.  PIN_PARITY_KEEP=1 keeps the temp dir; PIN_PARITY_REHEARSAL=1 skips what needs a GPU.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${REF:-/root/reference}
say() { echo "[pin_parity] $*"; }
die() { echo "[pin_parity] FAILED: $*" >&2; exit 1; }
command -v go > /dev/null || die "no Go toolchain on PATH (go: command not found) -- this is the script for the machine that has one"
[ -f "$REF/mpi.go" ] && [ -f "$REF/network.go" ] || die "REF=$REF is not a checkout of btracey/mpi (mpi.go / network.go missing)"
T=$(mktemp -d /tmp/pin_parity.XXXXXX) || die "mktemp"
[ -n "${PIN_PARITY_KEEP:-}" ] || trap 'rm -rf "$T"' EXIT
say "go: $(go version 2>/dev/null | head -1); reference: $REF (read only); work: $T"

# ---- 1. the reference as a module, with the collectives file in place
cp -r "$REF" "$T/ref" && chmod -R u+w "$T/ref" || die "copy of the reference"
rm -f "$T/ref/go.mod" "$T/ref/go.sum"
(cd "$T/ref" && go mod init github.com/btracey/mpi) || die "go mod init in the copy of the reference"
sed '1{/^\/\/go:build/d}' "$ROOT/go/mpi_collectives/collectives.go" > "$T/ref/collectives.go" || die "collectives.go"
grep -q '^//go:build' "$T/ref/collectives.go" && die "collectives.go still carries its build tag"

# ---- 2. this repository's Go module against it
cp -r "$ROOT/go" "$T/go" || die "copy of go/"
(cd "$T/go" && go mod edit -replace "github.com/btracey/mpi=$T/ref") || die "go mod edit -replace"
[ -f "$ROOT/mpi_amd/libxmpi.so" ] || (cd "$ROOT" && python -m mpi_amd.build) || die "python -m mpi_amd.build"
export CGO_CFLAGS="-I$ROOT/include ${CGO_CFLAGS:-}"
export CGO_LDFLAGS="-L$ROOT/mpi_amd -lxmpi -Wl,-rpath,$ROOT/mpi_amd ${CGO_LDFLAGS:-}"
export LD_LIBRARY_PATH="$ROOT/mpi_amd${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH}"
(cd "$T/ref" && go vet . && go build .) || die "the reference package with collectives.go dropped in does not vet / build"
(cd "$T/go" && go vet ./... && go build ./...) || die "go vet / go build of go/ (the cgo shim, the golden generator)"
say "acceptance 1: go vet ./... && go build ./... -- ok"

# ---- 3. the reference's own bytes and transcripts -> tests/golden, and the restatements held to them
(cd "$T/go" && go run ./golden -out "$ROOT/tests/golden") || die "go run ./golden"
(cd "$ROOT" && python -m pytest tests/test_reference_golden.py -q -rs) || die "tests/test_reference_golden.py: a restatement differs from the reference"
if ls "$ROOT"/tests/golden/ref_*.json > /dev/null 2>&1; then say "parity PINNED: tests/golden/ref_*.json written by the reference, the oracle and the product codec match them -- commit the fixtures"
else say "no fixtures were written (a rehearsal?): parity stays unpinned"; fi

# ---- 4. the shim on the GPU
gpu=no; [ -e /dev/kfd ] && [ -z "${PIN_PARITY_REHEARSAL:-}" ] && gpu=yes
if [ $gpu = yes ]; then (cd "$T/go" && go test ./xgmi) || die "go test ./xgmi"; say "go test ./xgmi -- ok"
else say "no GPU visible here (or a rehearsal): go test ./xgmi skipped -- run it on the GPU box"; fi

# ---- 5. the reference's two programs, their source files untouched, on libxmpi.so
mkdir -p "$T/run/helloworld" "$T/run/bounce"
cp "$REF/examples/helloworld/helloworld.go" "$T/run/helloworld/" && cp "$REF/examples/bounce/bounce.go" "$T/run/bounce/" || die "copy of the examples"
cmp -s "$REF/examples/helloworld/helloworld.go" "$T/run/helloworld/helloworld.go" || die "helloworld.go is not byte for byte the reference's"
for p in helloworld bounce; do
cat > "$T/run/$p/register_xgmi.go" <<GO
package main

import (
	"github.com/btracey/mpi"
	"github.com/btracey/mpi-xgmi/xgmi"
)

func init() { mpi.Register(&xgmi.Backend{}) } // mpi.go:61-67: the one line that puts the program on the MI355X backend
GO
done
cat > "$T/run/go.mod" <<MOD
module xmpi.local/run

go 1.18

require (
	github.com/btracey/mpi v0.0.0
	github.com/btracey/mpi-xgmi v0.0.0
)

replace github.com/btracey/mpi => $T/ref

replace github.com/btracey/mpi-xgmi => $T/go
MOD
(cd "$T/run" && go build -o "$T/run/helloworld.bin" ./helloworld) || die "the reference's helloworld.go + the one-line Register does not build"
built_bounce=yes
(cd "$T/run" && go build -o "$T/run/bounce.bin" ./bounce) || { built_bounce=no; say "bounce.go imports github.com/gonum/floats (bounce.go:29, one call: floats.Equal): not in the module cache and not fetchable here -- bounce skipped; helloworld stands"; }
say "acceptance 2a: the reference's examples build against the shim (helloworld: yes, bounce: $built_bounce)"
if [ $gpu = yes ]; then
  XMPI_BASEPORT=${XMPI_BASEPORT:-6400} "$ROOT/mpi_amd/bin/xmpirun" 2 "$T/run/helloworld.bin" || die "helloworld under xmpirun"
  [ $built_bounce = yes ] && { XMPI_BASEPORT=$(( ${XMPI_BASEPORT:-6400} + 20 )) "$ROOT/mpi_amd/bin/xmpirun" 2 "$T/run/bounce.bin" || die "bounce under xmpirun"; }
  say "acceptance 2b: the reference's programs ran on libxmpi.so through mpi.Register -- ok"
else say "no GPU visible here (or a rehearsal): the programs were built, not run"; fi
say "done"
