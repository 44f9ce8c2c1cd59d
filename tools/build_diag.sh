#!/bin/bash
# Diagnostics builds of libcharon_hip.so for A/B experiments (never shipped, never loaded by default):
#   tools/diag/libcharon_hip_fake_emit.so   -DCHN_K1_FAKE_EMIT: k_minimise_probe without hashing (probe pipeline alone)
# (CHN_DIAG_NO_STORE / CHN_DIAG_NO_ESC_STORE builds are NOT offered here any more: they leave the row log unwritten, and the
#  count kernel then chases garbage escaped-row indices -- a GPU memory fault; they were only ever meaningful with the
#  count kernel's output ignored.)
# Use with CHARON_HIP_LIB=<path> python bench.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/tools/diag
for v in "$@"; do
  case $v in
    fake_emit) D="-DCHN_K1_FAKE_EMIT" ;;
    fake_nobase) D="-DCHN_K1_FAKE_EMIT -DCHN_DIAG_NO_BASE" ;;
    nofast) D="-DCHN_K1_NO_FAST_STEP" ;;
    batch1) D="-DCHN_LOG_BATCH=1" ;;
    batch8) D="-DCHN_LOG_BATCH=8" ;;
    batch4_pieces8) D="-DCHN_LOG_BATCH=4 -DCHN_BASE_PIECES=8" ;;
    fake_batch4) D="-DCHN_K1_FAKE_EMIT -DCHN_LOG_BATCH=4" ;;
    pieces1) D="-DCHN_BASE_PIECES=1" ;;
    pieces2) D="-DCHN_BASE_PIECES=2" ;;
    pieces4) D="-DCHN_BASE_PIECES=4" ;;
    pieces8) D="-DCHN_BASE_PIECES=8" ;;
    allprobes) D="-DK1_COND_MAX_BINS=0" ;;
    condall) D="-DK1_COND_MAX_BINS=255" ;;
    k2big) D="-DK2_LDS_BUDGET=65536u" ;;
    *) D="-D$v" ;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++14 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -I$R/include -I$R/charon_amd/csrc $D -shared \
      -o $R/tools/diag/libcharon_hip_$v.so $R/charon_amd/csrc/charon_hip.hip
done
