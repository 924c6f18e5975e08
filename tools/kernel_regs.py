"""Print VGPR/AGPR/SGPR/scratch/LDS per kernel from a hipcc -save-temps .s file."""
import re, subprocess, sys
s = open(sys.argv[1]).read()
md = s[s.find('amdhsa.kernels'):]
blocks = md.split('- .agpr_count:')[1:]
for b in blocks:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
    try:
        d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except Exception:
        d = name
    d = d.replace('(anonymous namespace)::', '').replace('void ', '')
    ag = re.match(r'\s*(\d+)', b).group(1)
    print(f'{d[:64]:64s} vgpr={g("vgpr_count"):>3s} agpr={ag:>3s} sgpr={g("sgpr_count"):>3s} scratch={g("private_segment_fixed_size")} lds={g("group_segment_fixed_size")}')
