#!/bin/bash
# instruction-run summary of one kernel of ast_amd/_obj/<obj>.o: scratch/isa_runs.sh decoder_persist 'bwdILi8ELi3ELb0E'
# (runs of MFMAs, 16-byte buffer loads, scratch loads / stores, barriers: shows spill reloads sitting between the loads of a product)
bash scratch/kregs.sh $1 >/dev/null
/opt/rocm/lib/llvm/bin/llvm-objdump -d /tmp/kregs_$1.co | awk -v pat="$2" '/^[0-9a-f]+ <.*>:/{p = ($0 ~ pat)} p' > /tmp/isa_$1.s
grep -n "scratch_\|s_barrier\|v_mfma\|buffer_load_dwordx4" /tmp/isa_$1.s | awk '{k=$2; if (k ~ /v_mfma/) k="mfma"; if (k ~ /scratch_load/) k="SLOAD"; if (k ~ /scratch_store/) k="SSTORE"; if (k ~ /buffer_load/) k="bl4"; if (k!=last) { if (last!="") printf "%s %s x%d\n", first, last, n; first=$1; last=k; n=0 } n++ } END {printf "%s %s x%d\n", first, last, n}'
