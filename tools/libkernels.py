"""Which kernels did the vendor library pick (tools/bench_vs_library.py under rocprofv3 --kernel-trace)?  Full kernel name, workgroup
size, LDS bytes, VGPR/AGPR counts, grid and median duration per (kernel, grid): python tools/libkernels.py <dir>"""
import csv, glob, sys, statistics as st
g = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("Cijk") or "gemm" in r["Kernel_Name"].lower():
            k = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])
            g.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print(f"{st.median(v):9.1f} us n={len(v):4d} grid=({k[1]},{k[2]}) wg={k[3]} lds={k[4]} vgpr={k[5]} agpr={k[6]}\n      {k[0][:600]}")
