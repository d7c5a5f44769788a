#!/bin/bash
# register / spill / LDS figures of every kernel in one object of ast_amd/_obj: scratch/kregs.sh lstm_persist [name filter]
O=ast_amd/_obj/$1.o
objcopy -O binary --only-section=.hip_fatbin $O /tmp/kregs_$1.fb
T=$(/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=/tmp/kregs_$1.fb | grep gfx950)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=$T --input=/tmp/kregs_$1.fb --output=/tmp/kregs_$1.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kregs_$1.co | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    dn=re.sub(r'astk::\(anonymous namespace\)::','',dn)
    if len(sys.argv)>1 and sys.argv[1] not in dn: continue
    print(f\"{dn[:100]:100s} agpr {blk.split()[0]:>4s} vgpr {g('vgpr_count'):>4s} vspill {g('vgpr_spill_count'):>4s} sgpr {g('sgpr_count'):>4s} sspill {g('sgpr_spill_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}\")
" $2
