#!/bin/bash
# The GPU suite's collective tests (tests/test_gpu_collectives.py, everything but the full-size configs) with every rank process on a
# VIRTUAL device of its own (tests/devsim; no GPU) -- the layout an 8-GPU node gives them and this round's boxes cannot: rank i on
# device i, peer access, cross-device IPC.  ~6 min on 8 cores.  The CPU suite runs a chosen subset of this (tests/test_devsim.py).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
python -m tests.devsim.build || exit 1
PYTHONPATH=$ROOT/tests/devsim/site XMPI_DEVSIM_LIB=$ROOT/tests/devsim/libxmpi_devsim.so XMPI_TIMEOUT_S=120 \
  python -m pytest tests/test_gpu_collectives.py -m gpu -q --timeout 900 -k "not cfg3 and not cfg4 and not cfg5 and not coloured" "$@"
