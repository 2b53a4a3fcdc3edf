"""gaps between consecutive kernels of each queue in a rocprofv3 kernel_trace.csv (tools/gaps.sh)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n[:n.index('(')] if '(' in n else n
byq = collections.defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for q, ks in byq.items():
    ks.sort()
    ks = ks[len(ks) // 2:]                      # second repetition only (warm)
    for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
        dur[n0].append(e0 - s0)
        if n0.startswith(('k_rowgemm_h2', 'k_edge_h2', 'k_node')) and n1.startswith(('k_rowgemm_h2', 'k_edge_h2', 'k_node')):
            gap[(n0[:28], n1[:28])].append(s1 - e0)
import statistics as st
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print('  %-44s n %5d  median %7.2f us  mean %7.2f' % (n[:44], len(v), st.median(v) / 1e3, st.mean(v) / 1e3))
for k, v in gap.items():
    print('  gap %-28s -> %-28s n %5d  median %6.2f us  p10 %6.2f  p90 %6.2f' % (k[0], k[1], len(v), st.median(v) / 1e3, sorted(v)[len(v) // 10] / 1e3, sorted(v)[9 * len(v) // 10] / 1e3))
tot = {q: (ks[-1][1] - ks[len(ks) // 2][0]) / 1e3 for q, ks in ((q, sorted(k)) for q, k in byq.items()) if len(ks) > 20}
print('  span of the second repetition per queue (us):', {q: round(v, 1) for q, v in tot.items()})
