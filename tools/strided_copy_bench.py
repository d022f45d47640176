import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from cfpnet_amd import hip, ops
from _gtime import graph_time_us
hip.load()
DEV="cuda:0"
rows, C = 8*30*40, 816
NB=6
xs=[ops.Act(torch.randn(rows, C, device=DEV).to(torch.bfloat16),0,C) for _ in range(NB)]
ys=[ops.new_act(rows, C, torch.bfloat16, DEV) for _ in range(NB)]
k=[0]
for width in (816, 408, 128, 64):
    nsl = C // width
    def run():
        i=k[0]%NB; k[0]+=1
        for s in range(nsl):
            ops.copy_rows(xs[i].slice(s*width, width), ys[i].slice(s*width, width), rows)
    t = graph_time_us(run, calls=6, replays=4)
    print(f"copy {rows}x{C} as {nsl} column slices of {width} ch ({width*2} B runs): {t:7.1f} us total, {t/nsl:6.2f} us per slice launch")
