#!/bin/bash
# Print code size / VGPRs / scratch (spill) bytes of every kernel in the HIP sources: a kernel that starts spilling
# (ScratchSize > 0) after an edit can lose 2x without failing any test.
#   tools/kernel_resources.sh [file.hip ...]
cd "$(dirname "$0")/.."
files=("$@"); [ ${#files[@]} -eq 0 ] && files=(theia_amd/csrc/*.hip)
for f in "${files[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S "$f" -o /tmp/_kr.s 2>/dev/null || { echo "$f: compile failed"; continue; }
  awk -v F="$(basename "$f")" '/^; Kernel info:/ {k=1} /^\s*\.amdhsa_kernel / {name=$2} /codeLenInByte/ {code=$4} /; NumVgprs:/ {v=$3} /; NumAgprs:/ {a=$3} /; ScratchSize:/ {s=$3} /; Occupancy:/ {printf "%-22s code %7d B  vgpr %3d  agpr %3d  scratch %4d  occ %d  %s\n", F, code, v, a, s, $3, name}' /tmp/_kr.s
done
