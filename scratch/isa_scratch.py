"""where a kernel's scratch (spill) operations sit relative to its MFMAs and barriers: python3 scratch/isa_scratch.py <unit> <mangled-name filter>"""
import re, subprocess, sys, collections
unit, filt = sys.argv[1], sys.argv[2]
subprocess.run(["bash", "scratch/kregs.sh", unit, "zzzz"], capture_output=True)
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f"/tmp/kregs_{unit}.co"], capture_output=True, text=True).stdout
for f in re.split(r'\n(?=[0-9a-f]+ <)', txt):
    m = re.match(r'[0-9a-f]+ <(\S+)>', f)
    if not m or filt not in m.group(1): continue
    lines = f.split('\n')
    idx = [i for i, l in enumerate(lines) if 'scratch_' in l]
    mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    bar = [i for i, l in enumerate(lines) if 's_barrier' in l]
    print(m.group(1), len(lines), 'lines;', len(idx), 'scratch ops;', len(mf), 'mfma in lines', (mf[0], mf[-1]) if mf else None)
    print(' scratch ops inside the MFMA range:', sum(1 for i in idx if mf and mf[0] <= i <= mf[-1]), ' barriers at', bar[:40])
    print(' scratch lines:', idx)
    if len(sys.argv) > 3:
        a, b = map(int, sys.argv[3].split(':'))
        for l in lines[a:b]: print(l.split('//')[0].rstrip()[:120])
