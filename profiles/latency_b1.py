"""Latency of one controller instance (the drop-in case: MPC.get_control for a single car): mpmpc_solve with
B = 1 from host buffers, and the two kernels by HIP events.  python profiles/latency_b1.py  (on the GPU box)"""
import sys, time, numpy as np
sys.path[:0]=["multi-purpose-mpc_amd","tests","oracle","."]
import mpmpc, scenarios
tr=scenarios.sim_track(); sc=scenarios.make(2,tr,B=1)
Q,R,QN=scenarios.WEIGHTS["stock"]
cfg=mpmpc.make_config(sc.N,Q,R,QN,scenarios.XMIN,scenarios.XMAX,scenarios.UMIN,scenarios.UMAX,scenarios.AY_MAX,scenarios.CAR_LENGTH,max_batch=1)
h=mpmpc.Handle(cfg); h.set_path(tr.kappa,tr.v_ref,tr.ds_next)
for _ in range(20): h.solve(sc.wp_id,sc.x0,sc.cc_prev,sc.lb,sc.ub)
t=time.perf_counter()
for _ in range(500): h.solve(sc.wp_id,sc.x0,sc.cc_prev,sc.lb,sc.ub)
print("mpmpc_solve, B=1, host buffers: %.3f ms per call"%((time.perf_counter()-t)/500*1e3))
h.upload(sc.wp_id,sc.x0,sc.cc_prev,sc.lb,sc.ub)
a,s=zip(*[h.solve_resident_timed(1) for _ in range(50)])
print("K1 %.1f us, K2 %.1f us (events)"%(np.mean(a)*1e3,np.mean(s)*1e3))
