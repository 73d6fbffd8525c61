import os, subprocess, sys, time, tempfile
sys.path.insert(0, os.getcwd())
from mtr_amd import synth
reads=[c for _,c in synth.make_reads("headline2k", 10000, 2)]
td=tempfile.mkdtemp(); fa=os.path.join(td,"r.fa")
synth.write_fasta(fa, [(str(i), reads[i % len(reads)]) for i in range(100000)])
exe="mtr_amd/host/mTR"
for args, env in ((["-c"], {}), (["-c","-g","1"], {}), (["-c","-g","1"], {"MTR_GATHER":"rccl","MTR_GATHER_SELF":"1"}), (["-c","-g","2"], {})):
    e=dict(os.environ, MTR_HOST_TIMING="1", GPU_MAX_HW_QUEUES="8", **env)
    t0=time.perf_counter(); p=subprocess.run([exe,*args,fa],stdout=subprocess.DEVNULL,stderr=subprocess.PIPE,env=e); dt=time.perf_counter()-t0
    print("====", args, env, "%.3f s" % dt, "rc", p.returncode)
    print(p.stderr.decode()[-1800:])
