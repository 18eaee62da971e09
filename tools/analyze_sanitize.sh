#!/bin/bash
# Host analysis (ordering, elimination tree, supernodes, front rows) under ThreadSanitizer and AddressSanitizer + UBSan, CPU only:
#   bash tools/analyze_sanitize.sh [grid side, default 400]
set -e
cd "$(dirname "$0")/../kvxopt_amd/csrc"
G=${1:-400}
for san in thread address,undefined; do
  g++ -std=c++17 -O1 -g -fsanitize=$san -pthread -I. ../../tools/analyze_sanitize.cpp symbolic.cpp ordering.cpp amd_order.cpp -o /tmp/kvx_an_san
  echo "== -fsanitize=$san"
  KVX_ND_THREADS=8 KVX_ANALYZE_THREADS=8 /tmp/kvx_an_san $G
done
