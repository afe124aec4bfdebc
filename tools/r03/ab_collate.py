import os, sys, json, subprocess
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for rep in range(2):
    for fused in (0, 1024):
        code = f"import sys; sys.path.insert(0,'{root}'); import gae_dgl_amd.capture as c; c.FUSED_COLLATE_MAX_GRAPHS={fused}; sys.argv=['bench.py','--workload','zinc','--batch-graphs','128','--steps','300','--warmup','30','--no-extra','--no-cpu-baseline']; import runpy; runpy.run_path('{root}/bench.py', run_name='__main__')"
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout.strip().splitlines()
        l = json.loads([x for x in out if x.startswith("{")][-1])
        t = l["timing"]
        print("fused", fused, "median", round(t["ms_per_step_median"]*1e3,2), "min", round(t["ms_per_step_min"]*1e3,2), "max", round(t["ms_per_step_max"]*1e3,2), flush=True)
