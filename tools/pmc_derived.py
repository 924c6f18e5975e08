"""Derived per-kernel utilisation from tools/pmc_sq.sh's table (gpurun_out/pmc_sq.txt):
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   (four SIMDs per CU; 1.0 = every SIMD's MFMA pipe busy
               whenever its CU has work) -- the 'MFMA utilisation against gfx950 peak' of the conv implicit GEMMs
  lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE            (share of LDS index cycles lost to bank conflicts)
  valu_active  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES
Rows are one (kernel, grid size) each, ordered by CU-busy cycles."""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
hdr = lines[0].split()[1:]
col = {n: i for i, n in enumerate(hdr)}
def find(sfx):
    for n, i in col.items():
        if sfx.endswith(n) or n.endswith(sfx[-len(n):]):
            return i
    return None
iM, iB, iC, iI, iV, iW = (find(s) for s in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'SQ_LDS_BANK_CONFLICT',
                                              'SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES'))
print(f"{'kernel':58s} {'mfma_busy':>10s} {'lds_conflict':>13s} {'valu_active':>12s}")
for ln in lines[1:]:
    m = re.match(r'(.*? g=\d+)\s+(.*)$', ln)
    if not m:
        continue
    v = [float(t) for t in m.group(2).split()]
    if len(v) < len(hdr):
        continue
    mf = v[iM] / (4 * v[iB]) if v[iB] else 0
    lc = v[iC] / v[iI] if v[iI] else 0
    va = v[iV] / v[iW] if v[iW] else 0
    print(f"{m.group(1)[:58]:58s} {mf:10.3f} {lc:13.3f} {va:12.3f}")
