#!/bin/bash
# Build a copy of the library with -DKVX_PHASE_TIMING (cycle stamps along the path of the workgroup that updates tile (0, 0) and
# factors the next diagonal block) into scratch/libkvxhip_phase.so; run on the GPU box:  python scratch/phase_timing.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/kvx_phase_build && mkdir -p /tmp/kvx_phase_build/pkg/csrc && cp $ROOT/kvxopt_amd/csrc/*.hip $ROOT/kvxopt_amd/csrc/*.cpp $ROOT/kvxopt_amd/csrc/*.hpp $ROOT/kvxopt_amd/csrc/Makefile /tmp/kvx_phase_build/pkg/csrc/
mkdir -p /tmp/kvx_phase_build/include && cp $ROOT/include/*.h /tmp/kvx_phase_build/include/
make -s -j4 -C /tmp/kvx_phase_build/pkg/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DKVX_PHASE_TIMING"
cp /tmp/kvx_phase_build/pkg/libkvxhip.so $ROOT/scratch/libkvxhip_phase.so
ls -la $ROOT/scratch/libkvxhip_phase.so
