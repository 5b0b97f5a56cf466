"""Prints the per-kernel time table and the SQ counter ratios of one tools/gpu_round.sh visit.  Usage: show_round.py <tag> [steps in the stats run]"""
import csv, glob, json, sys
tag = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
f = glob.glob(f'gpurun_out/{tag}_stats/*/*_kernel_stats.csv')
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:16]:
        print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e6/steps:7.2f} ms/step {float(r['AverageNs'])/1e3:8.1f} us")
    print('total/step', round(tot / 1e6 / steps, 2))
try:
    A = json.load(open(f'gpurun_out/{tag}_pmcA.json')); B = json.load(open(f'gpurun_out/{tag}_pmcB.json'))
except FileNotFoundError:
    sys.exit(0)
for k in A:
    a, b = A[k], B.get(k, {})
    if a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) < 1e6: continue
    g = a['GRBM_GUI_ACTIVE'] / 8
    print(k[:60].ljust(60), f"mfma_util {a['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*g):.3f}  valu/mfma {b.get('SQ_INSTS_VALU',0)/max(b.get('SQ_INSTS_MFMA',1),1):.2f}  lds/mfma {b.get('SQ_INSTS_LDS',0)/max(b.get('SQ_INSTS_MFMA',1),1):.2f}"
          f"  wait_inst {a['SQ_WAIT_INST_ANY']/a['SQ_WAVE_CYCLES']:.2f}  wait_lds {a['SQ_WAIT_INST_LDS']/a['SQ_WAVE_CYCLES']:.2f}  valu_busy {4*b.get('SQ_ACTIVE_INST_VALU',0)/(1024*g):.2f}")
