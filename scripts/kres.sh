#!/bin/bash
# dev: per-kernel register / scratch / LDS summary of a HIP source (usage: scripts/kres.sh file.hip [filter])
cd "$(dirname "$0")/../mythos_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -fno-math-errno -fno-hip-fp32-correctly-rounded-divide-sqrt \
  -fgpu-flush-denormals-to-zero -fno-slp-vectorize --cuda-device-only $KRES_FLAGS -S -o /tmp/kres.s "$1" 2>/dev/null
awk '/^_Z[A-Za-z0-9_]*:/ {name=$1} /^; (ScratchSize|NumVgprs:|Occupancy|LDSByteSize)/ {printf "%s %s %s | ", substr(name,1,48), $2, $3; if ($2=="Occupancy:") printf "\n"}' /tmp/kres.s | grep "${2:-.}"
