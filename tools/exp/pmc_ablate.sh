set -u
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/pmc_ablate
mkdir -p $OUT
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES"; do
  n=$(echo $set | cut -c1-12 | tr ' ' '_')
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $OUT/$n -o p -- $OLDPWD/tools/exp/x6p_ablate > $OUT/$n.txt 2>&1)
done
python - <<'PY'
import sqlite3, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for db in glob.glob("gpurun_out/pmc_ablate/*/*.db"):
    c = sqlite3.connect(db)
    names = {r[0]: r[1] for r in c.execute("select dispatch_id, name from kernels")}
    grids = {r[0]: r[1] for r in c.execute("select dispatch_id, grid_x from kernels")}
    for did, cn, v in c.execute("select dispatch_id, counter_name, value from counters_collection"):
        k = names[did].split("gemm_x6p_kernel")[-1][:24] + f" g{grids[did]}"
        agg[k][cn].append(v)
for k in sorted(agg):
    if "<" not in k: continue
    d = {cn: sum(v) / len(v) for cn, v in agg[k].items()}
    gui = d.get("GRBM_GUI_ACTIVE", 0) / 8
    line = f"{k:34s}"
    if gui:
        line += f" mfma_util {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024):.3f} waves/cu {4 * d.get('SQ_WAVE_CYCLES', 0) / (gui * 256):.1f}"
    wc = d.get("SQ_WAVE_CYCLES")
    for cn in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA"):
        if cn in d: line += f" {cn[3:]}={d[cn]:.3g}"
    print(line)
PY
