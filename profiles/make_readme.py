"""Regenerates the measured sections of profiles/README.md from profiles/<round>/ (bench lines, rocprofv3
kernel stats, PMC summary):  python profiles/make_readme.py r1"""
import csv
import json
import os
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"
HERE = os.path.dirname(os.path.abspath(__file__))
R = os.path.join(HERE, rnd) + "/"
d2, d3, d4, db = (json.load(open(R + "bench_%s.json" % c)) for c in ("cfg2", "cfg3", "cfg4", "cfg2_b65536"))
p = json.load(open(R + "pmc_summary.json"))


def avg_ns(fname, what):
    return [float(r["AverageNs"]) for r in csv.DictReader(open(R + fname)) if what in r["Name"]][0]


def pmc(section, kernel, counter):
    return [v for k, v in p[section].items() if kernel in k][0][counter]["mean"]


k2, k1 = avg_ns("bench_cfg2_kernel_stats.csv", "solve"), avg_ns("bench_cfg2_kernel_stats.csv", "assemble")
k2b, k1b = avg_ns("bench_cfg2_b65536_kernel_stats.csv", "solve"), avg_ns("bench_cfg2_b65536_kernel_stats.csv", "assemble")
f, w = pmc("pmc_fetch", "solve", "FETCH_SIZE"), pmc("pmc_write", "solve", "WRITE_SIZE")
fa, wa = pmc("pmc_fetch", "assemble", "FETCH_SIZE"), pmc("pmc_write", "assemble", "WRITE_SIZE")
fb, wb = pmc("pmc_fetch_b65536", "solve", "FETCH_SIZE"), pmc("pmc_write_b65536", "solve", "WRITE_SIZE")
fab, wab = pmc("pmc_fetch_b65536", "assemble", "FETCH_SIZE"), pmc("pmc_write_b65536", "assemble", "WRITE_SIZE")
waves = pmc("pmc_sq1", "solve", "SQ_WAVES")
valu, salu, lds = (pmc("pmc_sq1", "solve", c) / waves for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"))
act = pmc("pmc_sq2", "solve", "SQ_ACTIVE_INST_VALU") / pmc("pmc_sq1", "solve", "SQ_INSTS_VALU")
wait = pmc("pmc_sq2", "solve", "SQ_WAIT_ANY") / pmc("pmc_sq1", "solve", "SQ_WAVE_CYCLES")
hist = d2["iters"]["ipm_histogram"]
st = d2["stock_osqp_settings"]
counts4 = d4["status_counts"]

import csv as _csv
_what = {"mpmpc_free_segments_kernel": "K0a: free runs of every waypoint's rasterised border line (200 threads)",
         "mpmpc_corridor_select_kernel": "K0b: segment selection along the horizon for 200 start waypoints x 50 columns",
         "mpmpc_localise_kernel": "K3a: waypoint search + path-relative state, 1 024 cars",
         "mpmpc_solve_kernel": "K2 inside the closed loop (1 024 cars, cold start)",
         "mpmpc_advance_kernel": "K3b: plan hand-over (one thread per plan entry) + plant step, 1 024 cars",
         "mpmpc_speed_profile_wave_kernel": "K4: one wavefront per path (mean over 1-path and 1 024-path launches)"}
_rows = []
try:
    for _r in _csv.DictReader(open(os.path.join(HERE, rnd, "next_rows_kernel_stats.csv"))):
        _name = _r["Name"].replace("void ", "").split("(")[0].split("<")[0]
        if _name in _what:
            _rows.append("| `%s` | %.1f µs (%s launches) | %s |" % (_name, float(_r["AverageNs"]) / 1e3, _r["Calls"], _what[_name]))
    NEXT_ROWS = "\n".join(_rows)
    NEXT_TXT = "".join("    " + l for l in open(os.path.join(HERE, rnd, "next_rows.txt")))
except OSError:
    NEXT_ROWS, NEXT_TXT = "| - | - | not collected |", "    (not collected)"
txt = f'''# profiles — measured on MI355X (gfx950), ROCm 7.2; host 2× AMD EPYC 9575F, of which the container may use 16 CPUs

Round 1 (`profiles/{rnd}/`).  All from `bench.py` runs on a fresh 1-GPU box; JSON lines are the bench
output, `*_kernel_stats.csv` is `rocprofv3 --kernel-trace --stats` of the same command,
`pmc_summary.json` holds per-kernel means of separate `rocprofv3 --pmc` passes of the same command
(FETCH_SIZE and WRITE_SIZE each in its own pass; the SQ counters in two passes of 8).  `collect.sh` gathers,
`summarize.py` condenses, `make_readme.py` writes the tables below; `census.py` (instruction census → flop
model), `pmc_admm.sh` (PMC cost of one ADMM iteration), `ab.sh` (A/B of two builds on one box), `quick.sh`,
`phases.py` (per-wave phase clocks of a `-DMPMPC_PHASE_CLOCK` build: where K2's time goes, which waves are the
slowest; outputs in `r1/phases_*.txt`), `sweep.sh` / `sweep_parity.sh` (solver-setting sweeps through `bench.py --set`),
`gsel.sh` (lanes per instance against batch size), `latency_b1.py` / `latency_get_control.py` (single-car calls),
`cpu_scaling.py` (thread scaling of the CPU baseline), `micro/exec_half.hip` (instruction-cost micro-benchmark) and
`micro/rsq_cost.hip` (a `v_rsq_f64` seed costs what `cvt + v_rsq_f32 + cvt` costs, 31 ns per rsqrt + cubic step in a
dependent chain, 21 ns with four independent ones, i.e. ≈ 3 ns per FP64 instruction of a single wave; both reach
1.2 ulp) are the helpers used while tuning.

## Headline (config 2: B = 1024 independent poses, N = 30, stock weights, free corridor)

| quantity | value | source |
|---|---|---|
| QP solves / s (assembly + solve, inputs resident in HBM) | **{d2["value"]/1e6:.2f} M** | `bench_cfg2.json` `value` |
| ms per step (batch of 1024) | {d2["ms_per_step"]:.3f} | `ms_per_step` |
| `max|u − u_ref|`, u = (v₀, δ₀), over all 1024 instances | {d2["max_abs_u_minus_uref"]:.1e} (tolerance 1e-6) | `max_abs_u_minus_uref` |
| whole plan without the cost-free κ_(N−1), e_ψ,N | {d2["max_abs_plan_minus_ref"]:.1e} | `max_abs_plan_minus_ref` |
| status agreement with the oracle | {100*d2["status_agreement"]:.0f} % | `status_agreement` |
| K2 `mpmpc_solve_kernel<64,16>` average launch | {d2["roofline"]["avg_ms"]:.4f} ms (HIP events) / {k2/1e6:.4f} ms (rocprofv3) | `roofline.avg_ms`, `bench_cfg2_kernel_stats.csv` |
| K1 `mpmpc_assemble_kernel` average launch | {d2["roofline_assembly"]["avg_ms"]*1e3:.1f} µs (events, includes event overhead) / {k1/1e3:.1f} µs (rocprofv3) | same |
| CPU baseline: C port of the reference-equivalent path, {d2["cpu_baseline"]["cores"]} threads (the container's CPU quota) | {d2["cpu_baseline"]["value"]/1e3:.1f} k solves/s | `cpu_baseline`, `cpu_scaling.txt` |
| same step from host buffers (`mpmpc_solve`, PCIe-inclusive) | {d2["host_buffers"]["value"]/1e6:.2f} M solves/s | `host_buffers` |
| ADMM iterations; interior-point iterations mean / max (histogram) | {d2["iters"]["admm_max"]} (early polish); {d2["iters"]["ipm_mean"]:.1f} / {d2["iters"]["ipm_max"]} ({", ".join("%s: %d" % kv for kv in sorted(hist.items(), key=lambda kv: int(kv[0])))}) | `iters` |
| the device at OSQP's defaults (ε = 1e-3, no polish) vs the optimum | up to {st["max_abs_u_minus_uref"]:.2f} rad in δ₀ after {st["admm_iters_mean"]:.0f} ADMM iterations (mean) | `stock_osqp_settings` |

Rooflines for the dominant kernel K2 at this size:

* HBM (`roofline`): algorithmic {d2["roofline"]["algorithmic_bytes"]/d2["config"]["batch_per_gpu"]:,.0f} B/solve (read wp_id, x0, the previous plan and the corridor
  rows; write z, y, u, status, iterations, residuals — the solve launch assembles its QP in registers) →
  {d2["roofline"]["achieved"]:.1f} GB/s = **{100*d2["roofline"]["frac"]:.2f} % of 8 TB/s**.  K2 is not an HBM kernel (DESIGN.md §5); the
  number is reported because the contract asks for it.
  PMC traffic: FETCH_SIZE {f:,.0f} KB (×2, gfx950 correction) + WRITE_SIZE {w:,.0f} KB = {(2*f+w)*1024/1e6:.1f} MB per launch against
  {d2["roofline"]["algorithmic_bytes"]/1e6:.1f} MB algorithmic ({(2*f+w)*1024/d2["roofline"]["algorithmic_bytes"]:.2f}×; no scratch: what is left above 1× is the shared path tables, the code and the
  partial 64-byte lines at the ends of the output rows).
  History: 5× while the loops spilled and the outputs were lane-strided 8-byte stores (WRITE_SIZE 23.7 MB); 1.16× with
  LDS-staged output rows; the materialised stage-blocked QP (6.7 KB per solve each way) went when K1 was fused in.
* FP64 vector (`roofline_fp64`): {d2["roofline_fp64"]["flops_per_solve_mean"]/1e6:.2f} MFLOP useful per solve (instruction census of the emulation,
  `census.py`) → {d2["roofline_fp64"]["achieved"]:.1f} TFLOP/s = **{100*d2["roofline_fp64"]["frac"]:.1f} % of 78.6 TFLOP/s**; 31 of 64 lanes hold a stage, and during the serial
  sweeps one lane per chain does useful work, which is what bounds this figure.
* SQ counters (per wave, mean): {valu/1e3:.0f} k VALU instructions, {salu/1e3:.1f} k SALU, {lds/1e3:.1f} k LDS; SQ_ACTIVE_INST_VALU /
  SQ_INSTS_VALU = {act:.2f} quad-cycles: with one wave per SIMD every vector instruction, FP64 or not, costs
  4 cycles, so kernel time ≈ 4 cycles × dynamic instruction count of the slowest wave (+ {100*wait:.0f} % SQ_WAIT_ANY).
  One ADMM iteration: 901 VALU + 60 SALU instructions, 1 165 quad-cycles = 1.9 µs (`pmc_admm.sh`); a taken
  loop-back branch costs about 25 cycles (`micro/exec_half.hip`), which is why the sweeps take four steps per trip.

K1 at this size moves 8.6 MB in {k1/1e3:.1f} µs (launch-latency dominated): {8601600/k1:.0f} GB/s = {100*8601600/k1/8000:.0f} % of peak by rocprofv3
time, {100*d2["roofline_assembly"]["frac"]:.1f} % by event time.  PMC: FETCH_SIZE {fa:.0f} KB (×2 = {2*fa*1024/1e6:.2f} MB; algorithmic inputs 1.03 MB),
WRITE_SIZE {wa:,.0f} KB (the padded 27 × 1024 × 32 doubles exactly).  At B = 65 536 (`bench_cfg2_b65536*`)
K1 takes {k1b/1e3:.0f} µs (rocprofv3; {db["roofline_assembly"]["avg_ms"]*1e3:.0f} µs by events) for 551 MB = **{550502400/k1b/1e3:.1f} TB/s, {100*550502400/k1b/8000:.0f} % of the 8 TB/s peak** ({100*550502400/k1b/6300:.0f} % of
the 6.3 TB/s achievable); PMC traffic there is {(2*fab+wab)*1024/1e6:.0f} MB.

## Single instance (the drop-in case)

`latency_b1.py`: one `mpmpc_solve` call with B = 1 from host buffers takes 0.17 ms end to end (K1 7 µs, K2 104 µs by
events; the rest is the PCIe copies through pinned staging and the launch path; 0.27 ms before the staging) — the reference spends ≈ 24 ms per control step in
Python + OSQP (SURVEY §8a).  `latency_get_control.py`: the whole `MPC.get_control()` + `drive()` step of the
host class takes 4.7 ms with the corridor computed on the host like the reference and 0.52 ms with
`corridor="device"` (K0 table on the GPU, rebuilt when the map changes).

## Other configurations (single runs, `--steps 5`)

Config 2's distribution at B = 2048 (two instances per wave from there on): 7.1 M solves/s, K2 0.27 ms —
the headline batch of 1024 fills every SIMD with one wave but only 31 of its 64 lanes.

| config | B | N | solves/s | K2 ms | status counts | max|u−u_ref| | status agreement | CPU port solves/s (threads) |
|---|---|---|---|---|---|---|---|---|
| 2 | 65 536 | 30 | {db["value"]/1e6:.2f} M | {db["roofline"]["avg_ms"]:.2f} (`<32,16>`: two instances per wave) | all solved | – | – | – |
| 3 time-optimal | 4 096 | 50 | {d3["value"]/1e6:.2f} M | {d3["roofline"]["avg_ms"]:.2f} (`<64,32>`) | all solved | {d3["max_abs_u_minus_uref"]:.1e} | {100*d3["status_agreement"]:.0f} % | {d3["cpu_baseline"]["value"]/1e3:.1f} k ({d3["cpu_baseline"]["cores"]}) |
| 4 obstacles | 8 192 | 30 | {d4["value"]/1e3:.0f} k | {d4["roofline"]["avg_ms"]:.1f} (`<32,16>`) | {counts4.get("1", 0)} solved, {counts4.get("-3", 0)} primal infeasible, {counts4.get("2", 0)} inaccurate | {d4["max_abs_u_minus_uref"]:.1e} | {100*d4["status_agreement"]:.0f} % | {d4["cpu_baseline"]["value"]/1e3:.1f} k ({d4["cpu_baseline"]["cores"]}) |

At B = 65 536 K2's PMC traffic is {(2*fb+wb)*1024/1e6:.0f} MB per launch against {db["roofline"]["algorithmic_bytes"]/1e6:.0f} MB algorithmic ({(2*fb+wb)*1024/db["roofline"]["algorithmic_bytes"]:.2f}×; the packed
variants have no scratch left either).

Config 4's time is set by the slowest waves: instances the early polish cannot certify restart the full
OSQP iteration; infeasible ones need a median of 675 and up to 3 925 ADMM iterations before OSQP's
certificate fires (solved ones: median 100, max 450), and five reach `max_iter` = 4 000, i.e. at least
4 000 × 2.1 µs of strictly serial work however the rest is scheduled.

CPU baseline: the GPU box reports 256 hardware threads but its cgroup allows 16 CPUs (`cpu.max` = 1600000/100000);
`cpu_scaling.txt` shows linear scaling to 16 threads (1.5 k solves/s per thread) and collapse beyond, so the
baseline runs on the quota.

A bench line's `roofline.traffic` is read from the `pmc_summary.json` that is present when `bench.py` runs, so
it reflects the previous PMC collection of the same build (`collect.sh` is run twice per refresh).

## The "next" rows (SURVEY §8f): K0, K3, K4

`next_rows.py` runs them in one process; `r1/next_rows_kernel_stats.csv` is `rocprofv3 --kernel-trace --stats` of it,
`r1/next_rows.txt` its own timings (host calls):

{NEXT_TXT}

| kernel | average | what |
|---|---|---|
{NEXT_ROWS}

## History of the headline in this round (config 2, solves/s)

1.44 M first fused kernel → 1.94 M (recurrence matrices, zero scratch) → 2.25 M (early polish) → 2.63 M
(early polish after 15, coalesced output rows) → 3.42 M (twisted factorisation) → 3.62 M (slack reciprocals)
→ 3.9 M (conditional refinement, LDS-parked deltas, SGPR constants) → 4.2 M (sweeps four steps per loop trip)
→ 4.3 M (warm-start floor from the ADMM residual) → 4.5 M (early attempt on four Ruiz passes) → 4.6–5.0 M (one
cubic Newton step in rsqrt / rcp, FMA-folded factor step and slack arithmetic; box-to-box spread ±4 %)
→ 4.9–5.2 M (interior-point stage in the split layout) → 5.2 M (no iterative refinement of the directions,
residual-based exit of the active-set refinement) → 5.8 M (early attempt after one ADMM iteration) → 6.0 M (on two Ruiz passes) → 6.4–6.8 M (reductions through DPP / permlane swaps instead of ds_bpermute) → 6.7 M (active-set
regularisation 1e-10) → 6.85 M (early attempt computes only the primal residual it uses) → 7.1–7.6 M (assembly inside
the solve launch, QP fields in registers / LDS: no stage-blocked QP in memory, no scratch, 410 VGPRs).
'''
open(os.path.join(HERE, "README.md"), "w").write(txt)
print("profiles/README.md written")
