"""which memory instructions consume spilled (scratch-reloaded) registers: python3 scratch/spill_uses.py <object.o> <mangled-name filter>"""
import re, subprocess, sys
obj, filt = sys.argv[1], sys.argv[2]
subprocess.run(f"objcopy -O binary --only-section=.hip_fatbin {obj} /tmp/su.fb", shell=True, check=True)
t = subprocess.run("/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=/tmp/su.fb", shell=True, capture_output=True, text=True).stdout.split()
t = [x for x in t if "gfx950" in x][0]
subprocess.run(f"/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets={t} --input=/tmp/su.fb --output=/tmp/su.co", shell=True, check=True)
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", "/tmp/su.co"], capture_output=True, text=True).stdout
for f in re.split(r'\n(?=[0-9a-f]+ <)', txt):
    m = re.match(r'[0-9a-f]+ <(\S+)>', f)
    if not m or filt not in m.group(1): continue
    L = [l.split('//')[0].strip() for l in f.split('\n')]
    print(m.group(1)[:70], len(L), 'lines')
    for i, l in enumerate(L):
        if 'scratch_load' not in l: continue
        off = re.search(r'offset:(\d+)', l)
        # follow the data flow a few steps: registers written by the reload, then by instructions that read them
        regs = set()
        mm = re.search(r'scratch_load_\w+ (v\[(\d+):(\d+)\]|v(\d+))', l)
        if mm.group(4): regs = {int(mm.group(4))}
        else: regs = set(range(int(mm.group(2)), int(mm.group(3)) + 1))
        found = None
        for j in range(i + 1, min(i + 60, len(L))):
            ins = L[j]
            if not ins or ins.startswith('s_'): continue
            ops = ins.split(None, 1)
            if len(ops) < 2: continue
            args = ops[1]
            used = set()
            for a, b, c in re.findall(r'v\[(\d+):(\d+)\]|v(\d+)', args):
                used |= {int(c)} if c else set(range(int(a), int(b) + 1))
            is_mem = any(k in ops[0] for k in ('global_', 'buffer_', 'flat_', 'ds_'))
            if used & regs:
                if is_mem: found = (j - i, ins[:90]); break
                # destination = first operand
                d = re.match(r'(v\[(\d+):(\d+)\]|v(\d+))', args)
                if d:
                    regs |= {int(d.group(4))} if d.group(4) else set(range(int(d.group(2)), int(d.group(3)) + 1))
        print(f"  line {i:6d} scratch offset {off.group(1) if off else '0':>4s} -> {found}")
