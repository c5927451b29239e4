#!/bin/bash
# Resource notes (registers, spills, scratch) and a static instruction mix of the search kernels, from a device-only compile -- no GPU needed.
#   bash benchmarks/kernel_resources.sh [probe|all] [out-dir]      probe = only ctc_next_kernel<4,512,false> and <2,1024,false> (fast)
cd "$(dirname "$0")/.."
MODE=${1:-probe}; OUT=${2:-/tmp/kres}; mkdir -p $OUT
DEF=""; [ "$MODE" = probe ] && DEF="-DSTT_CTC_PROBE"
LLVM=/opt/rocm/lib/llvm/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result $DEF $EXTRA --cuda-device-only -c stt_amd/csrc/ctc.hip -o $OUT/ctc.bundle -Iinclude || exit 1
$LLVM/clang-offload-bundler --unbundle --type=o --input=$OUT/ctc.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$OUT/ctc.elf
$LLVM/llvm-readelf --notes $OUT/ctc.elf | grep -E "^\s+\.name:|sgpr_count|sgpr_spill|vgpr_count|vgpr_spill|private_segment_fixed" | paste - - - - - - | sed 's/ \+/ /g; s/_Z15ctc_next_kernelILi\([0-9]\)ELi\([0-9]*\)ELb\([01]\)E[^ \t]*/ctc_next<\1,\2,\3>/' | grep -E "ctc_next|ctc_decode"
$LLVM/llvm-objdump -d --no-show-raw-insn $OUT/ctc.elf > $OUT/ctc.s
python - "$OUT/ctc.s" <<'PY'
import re, sys, collections
cur=None; mix=collections.defaultdict(collections.Counter)
for ln in open(sys.argv[1]):
    m=re.match(r'^[0-9a-f]+ <(\S+)>:', ln)
    if m: cur=m.group(1); continue
    m=re.match(r'^\s+(\w+)', ln)
    if cur and m:
        op=m.group(1)
        k=('readlane/writelane' if op.startswith(('v_readlane','v_writelane')) else 'scratch' if op.startswith('scratch_') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') and not op.startswith(('s_load','s_waitcnt','s_branch','s_cbranch','s_barrier','s_nop','s_sleep','s_buffer')) else 'smem' if op.startswith(('s_load','s_buffer')) else 'branch' if op.startswith(('s_branch','s_cbranch')) else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_','flat_','buffer_')) else 'other')
        mix[cur][k]+=1
for k,v in mix.items():
    if 'ctc_next' in k and ('Li4ELi512' in k or 'Li2ELi1024ELb0' in k):
        print(k[:40], dict(v), 'total', sum(v.values()))
PY
