#!/bin/bash
# tools/isa.sh FILE.hip [KERNEL-NAME-REGEX] -- the gfx950 ISA of one source (device side only) into /tmp/isa/FILE.s; with a regex,
# prints the line range and the instruction-class counts of every matching kernel (s_waitcnt vmcnt(0), mfma, scratch ...).
set -e
src=$1; pat=${2:-}
mkdir -p /tmp/isa
base=$(basename ${src%.hip})
extra=$(grep '^// HIPCC_FLAGS:' $src | cut -d: -f2-)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $extra --cuda-device-only -S -o /tmp/isa/$base.s $src
[ -z "$pat" ] && exit 0
python3 - "$pat" /tmp/isa/$base.s <<'PY'
import re, sys
pat, path = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        start, name = i, m.group(1)
    if l.startswith("\t.end_amdhsa_kernel") or l.strip().startswith(".size") and start is not None and False:
        pass
    if l.strip().startswith("s_endpgm") and start is not None:
        pass
# kernels: from "name:" to ".Lfunc_end"
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if m and re.search(pat, m.group(1)):
        j = i
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i:j]
        cnt = lambda r: sum(1 for b in body if re.search(r, b))
        print("%s  lines %d-%d  insns %d | mfma %d  vmcnt(0) %d  waitcnt %d  scratch %d  global_load %d  buffer_load %d  ds_read %d  ds_write %d  branches %d" % (
            m.group(1)[:110], i + 1, j + 1, cnt(r"^\t[sv]_|^\t(global|buffer|ds|flat|scratch)_"), cnt(r"v_mfma"), cnt(r"vmcnt\(0\)"), cnt(r"s_waitcnt"),
            cnt(r"scratch_"), cnt(r"global_load"), cnt(r"buffer_load"), cnt(r"ds_read|ds_load"), cnt(r"ds_write|ds_store"), cnt(r"s_cbranch")))
        i = j
    i += 1
PY
# the kernel descriptors' register counts
grep -E "^\s+\.(sgpr|vgpr|agpr)_count|\.name:|vgpr_spill|\.private_segment_fixed_size" /tmp/isa/$base.s | paste - - - - - - 2>/dev/null | grep -E "${pat}" | sed 's/  */ /g' | cut -c1-260
