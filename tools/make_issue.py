#!/usr/bin/env python3
"""Folds the SQ counter passes of bench.py (tools/pmc_summary.py output) and the VALU issue microbenchmark
(tools/microbench/valu_issue.hip) into profiles/rNN_issue.json: the issue-side roofline of the search kernel that
bench.py quotes as `roofline_issue`.
usage: make_issue.py <pmc_fused.txt> <valu_issue.txt> <out.json> [kernel_substring]"""
import ast
import json
import re
import sys

pmc_path, mb_path, out_path = sys.argv[1:4]
want = sys.argv[4] if len(sys.argv) > 4 else "k_icp_fused_dense"
counters = {}
launches = 0
for line in open(pmc_path):
    if want not in line:
        continue
    m = re.search(r"(\{.*\}) launches (\d+)", line)
    if m:
        counters.update(ast.literal_eval(m.group(1)))
        launches = int(m.group(2))
# per-SIMD cost of one VALU wave-instruction in the kernel's own mix, 8 waves on the SIMD: the candidate-scoring
# sequences of icp_dense.hpp as the microbenchmark runs them
mix = {}
for line in open(mb_path):
    m = re.match(r"(CANDIDATE \S+) \((\d+) VALU\)\s+waves/SIMD 8: .*kernel [\d.]+ us = ([\d.]+) ns per inst per SIMD", line)
    if m:
        mix[m.group(1)] = float(m.group(3))          # ns per candidate per SIMD
single = {}
for line in open(mb_path):
    m = re.match(r"(\S+)\s+waves/SIMD 8: .*\(([\d.]+) ticks/ns\).*= ([\d.]+) ns per inst per SIMD", line)
    if m:
        single[m.group(1)] = {"ns_per_inst_per_simd": float(m.group(3)), "s_memtime_ticks_per_ns": float(m.group(2))}
# VALU instructions of the packed candidate as hipcc emits it in the microbenchmark: 9 of the arithmetic + on average
# 1.56 register moves (counted in the ISA); the scalar form: 10
ns_packed = mix.get("CANDIDATE packed+u64key", 0.0) / 10.56
ns_scalar = mix.get("CANDIDATE scalar+f32cmp", 0.0) / 10.0
ns_mix = ns_packed or ns_scalar
clock_ghz = 2.4
out = {
    "kernel": want, "launches_counted": launches, "counters_per_launch": counters,
    "valu_insts": counters.get("SQ_INSTS_VALU"), "salu_insts": counters.get("SQ_INSTS_SALU"), "vmem_insts": counters.get("SQ_INSTS_VMEM_RD"),
    "simds": 1024,
    "ns_per_valu_inst_per_simd": ns_mix, "cycles_per_inst_at_2p4ghz": ns_mix * clock_ghz,
    "cycles_per_inst_source": "tools/microbench/valu_issue.hip on MI355X, 8 waves per SIMD, the candidate-scoring instruction mix of "
                              "icp_dense.hpp (packed form %.2f ns, scalar form %.2f ns per VALU per SIMD); single kinds: %s"
                              % (ns_packed, ns_scalar, ", ".join("%s %.2f" % (k, v["ns_per_inst_per_simd"]) for k, v in sorted(single.items())
                                                                  if k in ("v_add_f32", "v_fma_f32", "v_pk_add_f32", "v_min_f32", "v_cmp_lt_u64", "v_add_f64"))),
    "single_instruction_kinds": single,
    "lane_utilisation": (counters.get("SQ_THREAD_CYCLES_VALU", 0) / counters["SQ_ACTIVE_INST_VALU"] / 64.0) if counters.get("SQ_ACTIVE_INST_VALU") else None,
    "wave_cycles_split": {k: counters.get(k) for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU")},
}
out["bound_us"] = out["valu_insts"] * ns_mix / out["simds"] * 1e-3 if out["valu_insts"] else None
# the vector memory side: a wave-instruction occupies the CU's address unit for (64 lanes x bytes per lane) / 64 B per
# clock whatever the number of active lanes -- 16 cycles for the 128-bit loads that are nine tenths of this kernel's
# (profiles/r03_what_bounds_the_kernel.txt: four more such loads per trip, by one lane or by all, cost the same +37 us)
out["cus"] = 256
out["vmem_cycles_per_inst"] = 16.0
out["vmem_cycles_source"] = ("128-bit buffer loads: 64 lanes x 16 B at the L1's 64 B / clock; confirmed by profiles/r03_what_bounds_the_kernel.txt "
                             "(+4 loads per trip = +37 us per launch, with one active lane as with all)")
out["vmem_bound_us"] = (counters.get("SQ_INSTS_VMEM_RD", 0) * out["vmem_cycles_per_inst"] / out["cus"] / 2.3e9) * 1e6 if counters.get("SQ_INSTS_VMEM_RD") else None
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("valu_insts", "ns_per_valu_inst_per_simd", "bound_us", "vmem_bound_us", "lane_utilisation")}))
