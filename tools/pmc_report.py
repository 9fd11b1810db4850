import csv, glob, sys
tag = sys.argv[1]
vals = {}
ms = None
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            vals[r["Counter_Name"]] = float(r["Counter_Value"])
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            meta = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
print(f"render_kernel {ms:.2f} ms grid/wg/lds/vgpr/sgpr/scratch = {meta}")
for k in sorted(vals):
    print(f"{k:30s} {vals[k]:.6g}")
g = vals.get
if g("GRBM_GUI_ACTIVE") and g("SQ_INSTS_VALU"):
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    print(f"clock ~ {cyc/ms/1e6:.2f} GHz; VALU wave-instructions per SIMD-cycle = {g('SQ_INSTS_VALU')/(cyc*1024):.3f} (peak 0.5)")
    if g("SQ_INSTS_MFMA"):
        print(f"MFMA wave-instructions per SIMD-cycle = {g('SQ_INSTS_MFMA')/(cyc*1024):.4f} (32x32x16 f16: 32 cycles each -> matrix pipe busy {g('SQ_INSTS_MFMA')*32/(cyc*1024):.3f})")
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CU_CYCLES"):
        print(f"SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES = {g('SQ_VALU_MFMA_BUSY_CYCLES')/g('SQ_BUSY_CU_CYCLES'):.3f}")
    if g("SQ_WAVE_CYCLES"):
        print(f"avg waves resident per SIMD = {g('SQ_WAVE_CYCLES')*4/(cyc*1024):.2f}")
