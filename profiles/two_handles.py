import sys, time
sys.path[:0]=['/root/repo/multi-purpose-mpc_amd','/root/repo']
import numpy as np, mpmpc, scenarios
tr=scenarios.sim_track()
for conf,B,steps in ((2,1024,400),(4,8192,60),(2,4096,100)):
    sc=scenarios.make(conf,tr,B=B)
    Q,R,QN=scenarios.WEIGHTS[sc.weights]
    cfg=mpmpc.make_config(sc.N,Q,R,QN,scenarios.XMIN,scenarios.XMAX,scenarios.UMIN,scenarios.UMAX,scenarios.AY_MAX,scenarios.CAR_LENGTH,max_batch=B)
    hs=[mpmpc.Handle(cfg) for _ in range(2)]
    for h in hs:
        h.set_path(tr.kappa,tr.v_ref,tr.ds_next); h.set_outputs(False); h.upload(sc.wp_id,sc.x0,sc.cc_prev,sc.lb,sc.ub)
        for _ in range(200): h.solve_resident(B)
        h.sync()
    for nh in (1,2):
        for rep in range(2):
            t0=time.perf_counter()
            for i in range(steps): hs[i%nh].solve_resident(B)
            for h in hs[:nh]: h.sync()
            dt=time.perf_counter()-t0
        print('cfg',conf,'B',B,'handles',nh,'%.2f M solves/s'%(steps*B/dt/1e6),'%.4f ms/step'%(dt/steps*1e3))

# ... and PCIe-inclusive: the staged host path (inputs written into the pinned block every call, results read there), one handle
# synchronously against two handles kept busy with staged_begin / staged_end
for conf, B, steps in ((2, 1024, 200), (4, 8192, 40)):
    sc = scenarios.make(conf, tr, B=B)
    Q, R, QN = scenarios.WEIGHTS[sc.weights]
    cfg = mpmpc.make_config(sc.N, Q, R, QN, scenarios.XMIN, scenarios.XMAX, scenarios.UMIN, scenarios.UMAX, scenarios.AY_MAX, scenarios.CAR_LENGTH, max_batch=B)
    hs = [mpmpc.Handle(cfg) for _ in range(2)]
    vs = []
    for h in hs:
        h.set_path(tr.kappa, tr.v_ref, tr.ds_next)
        vs.append(h.staging(B))

    def fill(v):
        v["wp_id"][:] = sc.wp_id; v["x0"][:] = sc.x0; v["cc_prev"][:] = sc.cc_prev; v["lb"][:] = sc.lb; v["ub"][:] = sc.ub

    for want_z in (True, False):
        for h, v in zip(hs, vs):
            for _ in range(20):
                fill(v); h.solve_staged(B, want_z=want_z)
        t0 = time.perf_counter()
        for i in range(steps):
            fill(vs[0]); hs[0].solve_staged(B, want_z=want_z)
        one = steps * B / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        fill(vs[0]); hs[0].staged_begin(B, want_z=want_z)
        for i in range(1, steps):
            fill(vs[i & 1]); hs[i & 1].staged_begin(B, want_z=want_z)      # the next batch goes up while the previous one is solved
            hs[(i - 1) & 1].staged_end()
            _ = vs[(i - 1) & 1]["u0"][0, 0]
        hs[(steps - 1) & 1].staged_end()
        two = steps * B / (time.perf_counter() - t0)
        print('cfg', conf, 'B', B, 'staged host path,', 'plan copied back' if want_z else 'controls only', ': one handle %.2f M solves/s, two handles %.2f M' % (one / 1e6, two / 1e6))
