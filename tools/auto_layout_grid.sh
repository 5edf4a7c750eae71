# Every kernel form against the automatic choice on a dense grid of batch sizes (1080p, both profiles): flags sizes at which
# pick_layout is more than 4 % off the best form.  usage (GPU box, repo root): bash tools/auto_layout_grid.sh [WxH in macroblocks] [sizes] [tag]
MBS=${1:-120x68}
S=${2:-1,2,3,4,8,16,32,48,64,68,80,96,113,128,150,173,200,256,294,308,350,400,512,640,768,860,900,1024,1100,1400,1700,1800,2048,2300,2560,3072,4096}
TAG=${3:-r04r_grid}
for prof in baseline high; do
  timeout -k 10 500 python tools/layout_crossover.py --profile $prof --mbs $MBS --sizes $S --layouts pipe,pipe1,pipe1:1,wide,quad_wide:4,quad_wide:8,quad,oct,auto 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_$prof.log
done
TAG=$TAG python - <<'PY'
import re
for prof in ("baseline","high"):
    import os
    L=open(f"gpurun_out/{os.environ.get('TAG','r04r_grid')}_{prof}.log").read().split("\n")
    names=L[0].split()[1:10]
    print(prof)
    for l in L[1:]:
        t=l.split()
        if not t: continue
        n=int(t[0]); vals=[]
        for x in t[1:]:
            m=re.match(r"([0-9.]+)\[",x)
            vals.append(float(m.group(1)) if m else None)
        vals=vals[:9]
        auto=vals[-1]; best=min(v for v in vals[:-1] if v); bi=[i for i,v in enumerate(vals[:-1]) if v==best][0]
        flag = "  <-- auto %.1f %% off (%s)"%((auto/best-1)*100,names[bi]) if auto>best*1.04 else ""
        print("%5d auto %.3f best %.3f %s%s"%(n,auto,best,names[bi],flag))
PY
