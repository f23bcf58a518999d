import sys,re,subprocess
src=sys.argv[1]
out=subprocess.run(f"hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-value -x hip $H2W_EXTRA -c {src} -o /tmp/ra.o -Rpass-analysis=kernel-resource-usage",shell=True,capture_output=True,text=True).stderr
cur=None;rows={}
for l in out.splitlines():
    m=re.search(r'remark: +(.*?): (.*?) \[-Rpass',l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=='Function Name':
        cur=subprocess.run(['c++filt',v],capture_output=True,text=True).stdout.strip()[:70]; rows[cur]={}
    elif cur: rows[cur][k]=v
for n,r in rows.items():
    print(f"{n:70s} S{r.get('TotalSGPRs'):>4} V{r.get('VGPRs'):>4} A{r.get('AGPRs'):>3} scr{r.get('ScratchSize [bytes/lane]'):>5} occ{r.get('Occupancy [waves/SIMD]'):>2} sspill{r.get('SGPRs Spill'):>3} vspill{r.get('VGPRs Spill'):>3} lds{r.get('LDS Size [bytes/block]'):>6}")
