"""Which kernels wait for their global loads one at a time?  Per kernel of the library: global / buffer loads, s_waitcnt vmcnt
instructions, and the longest run of loads with no wait in between (static order of the assembly: branches make it approximate,
which is enough to find loops of 'load, wait, use').  A kernel that hides such chains behind other waves when it runs alone pays
for every one of them beside the other streams' kernels, where a load takes several times as long.

  python tools/isa_waits.py [file.hip ...]          # default: every .hip of wesup_amd/csrc
"""
import glob, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(root, 'wesup_amd', 'csrc', '*.hip')))
print(f'{"kernel":70s} {"loads":>6s} {"waits":>6s} {"longest run":>12s} {"vgprs":>6s}')
for f in files:
    asm = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-gpu-rdc', '-S', '--cuda-device-only',
                          f, '-o', '-', '-I' + os.path.join(root, 'include')], capture_output=True, text=True).stdout
    vg = dict(re.findall(r'\.name:\s+(\S+)\n(?:.*\n){0,14}?\s+\.vgpr_count:\s+(\d+)', asm))
    cur, stats = None, {}
    for line in asm.split('\n'):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1); stats[cur] = [0, 0, 0, 0]          # loads, waits, run, best
            continue
        if cur is None:
            continue
        t = line.strip().split(' ')[0].split('\t')[0]
        if t.startswith('global_load') or (t.startswith('buffer_load') and ' lds' not in line):
            s = stats[cur]; s[0] += 1; s[2] += 1; s[3] = max(s[3], s[2])
        elif t == 's_waitcnt' and 'vmcnt' in line:
            s = stats[cur]; s[1] += 1; s[2] = 0
        elif t == 's_endpgm':
            cur = None
    for k, (l, w, _, best) in stats.items():
        if l >= 4:
            name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
            name = re.sub(r'\(.*', '', name).replace('void ', '')
            flag = '  <-- one at a time' if best <= 2 and l >= 8 else ''
            print(f'{os.path.basename(f)[:14]:14s} {name[:55]:55s} {l:6d} {w:6d} {best:12d} {vg.get(k, "?"):>6s}{flag}')
