"""Condense gpurun_out/<tag>/ of profiles/pmc_s2.sh (SQ counters of the pair kernel against the one-stage kernel):  python profiles/pmc_s2_show.py gpurun_out/s2pmc > profiles/r6/pmc_s2.txt"""
import csv, glob, collections, sys, os, re
O=sys.argv[1]
for name in sorted(set(re.sub(r"_(sq|mix|x|trace)$","",os.path.basename(d)) for d in glob.glob(O+"/*_sq"))):
    agg=collections.defaultdict(list)
    for grp in ("sq","mix","x"):
        for f in glob.glob("%s/%s_%s/*/*counter_collection.csv"%(O,name,grp)):
            for r in csv.DictReader(open(f)):
                if "mpmpc_reduced" in r["Kernel_Name"] and "tail" not in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c={k:sum(v)/len(v) for k,v in agg.items()}
    if not c: continue
    w=c["SQ_WAVES"]; wc=c["SQ_WAVE_CYCLES"]
    f64=sum(c.get(n,0) for n in ("SQ_INSTS_VALU_FMA_F64","SQ_INSTS_VALU_MUL_F64","SQ_INSTS_VALU_ADD_F64","SQ_INSTS_VALU_TRANS_F64"))
    print(name,"waves %d VALU/wave %.0f SALU/wave %.0f LDS/wave %.0f branch/wave %.0f smem/wave %.0f f64 %.0f%% wavecycles/wave %.0f  cyc/VALU %.2f"%(w,c["SQ_INSTS_VALU"]/w,c["SQ_INSTS_SALU"]/w,c["SQ_INSTS_LDS"]/w,c.get("SQ_INSTS_BRANCH",0)/w,c.get("SQ_INSTS_SMEM",0)/w,100*f64/c["SQ_INSTS_VALU"],wc/w, 4*wc/c["SQ_INSTS_VALU"]))
    print("    shares of wave cycles: "+", ".join("%s %.0f%%"%(n[3:],100*c[n]/wc) for n in ("SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_MISC","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_WAIT_INST_LDS") if n in c))
