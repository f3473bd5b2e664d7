import csv, collections, sys
path=sys.argv[1]
rows=list(csv.DictReader(open(path)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
naive=[i for i,r in enumerate(rows) if 'naive_conv' in r['Kernel_Name']]
rs=rows[(max(naive)+1 if naive else 0):]
adam=[i for i,r in enumerate(rs) if 'adam' in r['Kernel_Name'].lower()]
if not adam:      # other optimizers (GFL: SGD with momentum through ATen's multi-tensor kernels)
    adam=[i for i,r in enumerate(rs) if 'sgd' in r['Kernel_Name'].lower() or 'multi_tensor_apply' in r['Kernel_Name']]
groups=[]; prev=None
for i in adam:
    if prev is None or i-prev>50: groups.append([i])
    else: groups[-1].append(i)
    prev=i
a=groups[-2][-1]+1; b=groups[-1][-1]+1
step=rs[a:b]
st=int(step[0]['Start_Timestamp']); en=int(step[-1]['End_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in step)
print(f"last step: wall {(en-st)/1e6:.1f} ms, kernel busy {busy/1e6:.1f} ms, n kernels {len(step)}")
agg=collections.defaultdict(lambda:[0,0])
def cat(n):
    if 'msda' in n: return n[n.index('msda'):][:40]
    if n.startswith('Cijk') : return 'GEMM (hipBLASLt Cijk)'
    if 'conv' in n.lower() or 'igemm' in n.lower() or 'gemm_xdl' in n or 'ck::' in n: return 'conv/ck: '+n[:50]
    return n[:95]
for r in step:
    key=cat(r['Kernel_Name'])
    agg[key][0]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); agg[key][1]+=1
N=int(sys.argv[2]) if len(sys.argv)>2 else 40
for k,(t,c) in sorted(agg.items(), key=lambda kv:-kv[1][0])[:N]:
    print(f"{t/1e6:8.2f} ms {c:5d}  {k}")
# per-queue busy time / union coverage of the last step
qs = collections.defaultdict(list)
for r in step:
    qs[r.get('Queue_Id', r.get('Stream_Id', '?'))].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
allint = sorted((s_, e_) for v in qs.values() for s_, e_ in v)
cov = 0; cur_s, cur_e = allint[0]
for s_, e_ in allint[1:]:
    if s_ > cur_e: cov += cur_e - cur_s; cur_s, cur_e = s_, e_
    else: cur_e = max(cur_e, e_)
cov += cur_e - cur_s
print(f"union of kernel intervals {cov/1e6:.1f} ms of wall {(en-st)/1e6:.1f} ms -> GPU idle {(en-st-cov)/1e6:.1f} ms")
for q, v in qs.items():
    print(f"  queue {q}: {len(v)} kernels, busy {sum(e_-s_ for s_,e_ in v)/1e6:.1f} ms, span {(max(e_ for _,e_ in v)-min(s_ for s_,_ in v))/1e6:.1f} ms")
# hipBLASLt GEMMs of the step by (kernel, grid): count, average and total time
g = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r['Kernel_Name']
    if n.startswith('Cijk'):
        mt = n[n.index('_MT'):][:16] if '_MT' in n else ''
        key = (n[:24] + mt, r.get('Grid_Size', r.get('Grid_Size_X', '?')))
        g[key][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); g[key][1] += 1
print("GEMM groups (name, grid): total ms, count, avg us")
for k, (t, c) in sorted(g.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t/1e6:7.2f} ms {c:4d} {t/c/1e3:8.1f} us  {k[0]}  grid={k[1]}")
# the longest individual launches of ATen's generic (strided / casting) elementwise kernel
mu = [(int(r['End_Timestamp']) - int(r['Start_Timestamp']), r.get('Grid_Size', '?'), r['Kernel_Name'][60:200]) for r in step
      if 'manual_unroll' in r['Kernel_Name']]
mu.sort(reverse=True)
print("elementwise_kernel_manual_unroll, longest launches: us, grid, functor")
for t, gsz, n in mu[:30]:
    print(f"{t/1e3:8.1f} us  grid={gsz:>10s}  {n}")
# the longest individual ATen launches of the step (any kernel of at::native)
at = [(int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']) for r in step if 'at::native' in r['Kernel_Name']]
at.sort(reverse=True)
print("ATen kernels, longest launches: us, name (functor part)")
import re
for t, n in at[:60]:
    short = re.sub(r"at::native::|\(anonymous namespace\)::|std::array<char\*, \d+ul>|at::TensorIteratorBase&", "", n)
    print(f"{t/1e3:8.1f} us  {short[:170]}")
tot = sum(t for t, _ in at)
big = sum(t for t, _ in at if t >= 15000)
print(f"ATen total {tot/1e6:.2f} ms in {len(at)} launches; launches >= 15 us: {big/1e6:.2f} ms in {sum(1 for t,_ in at if t>=15000)}")
# ---- idle time of the busiest queue by the kernel that FOLLOWS the gap, and the gap-size histogram (r4): where the main
# stream's dependency gaps sit
mainq = max(qs.items(), key=lambda kv: len(kv[1]))[0]
mrows = [r for r in step if r.get('Queue_Id', r.get('Stream_Id', '?')) == mainq]
mrows.sort(key=lambda r: int(r['Start_Timestamp']))
gaps = collections.defaultdict(lambda: [0, 0])
hist = collections.Counter()
tot_gap = 0
pend = int(mrows[0]['End_Timestamp'])
for prev, r in zip(mrows, mrows[1:]):
    g_ = int(r['Start_Timestamp']) - pend
    pend = max(pend, int(r['End_Timestamp']))
    if g_ <= 0:
        continue
    tot_gap += g_
    key = cat(prev['Kernel_Name'])[:60] + '  ->  ' + cat(r['Kernel_Name'])[:60]
    gaps[key][0] += g_; gaps[key][1] += 1
    hist[min(int(g_ / 1000), 50)] += 1
print(f"queue {mainq}: {len(mrows)} launches, idle between them {tot_gap/1e6:.2f} ms; gap histogram (us: count): " +
      " ".join(f"{k}:{v}" for k, v in sorted(hist.items())))
print("largest idle contributions (previous kernel -> next kernel): total us, count, avg us")
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{t/1e3:8.1f} us {c:4d} {t/c/1e3:6.1f}  {k}")
