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
