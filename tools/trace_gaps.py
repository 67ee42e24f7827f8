"""Kernel-to-kernel gaps of the last Jive launches in a rocprofv3 --kernel-trace directory (argv[1])."""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the last run of k_jive / coop kernels (a merkle chain): take last 21 kernels whose name has k_jive
ks=[r for r in rows if 'k_jive' in r['Kernel_Name']]
print(len(ks),"jive kernels")
chain=ks[-21:]
tot=0
for a,b in zip(chain,chain[1:]):
    gap=(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3
    dur=(int(a['End_Timestamp'])-int(a['Start_Timestamp']))/1e3
    tot+=gap
    print("%-12s dur %9.1f us  gap to next %7.1f us  grid %s"%(('coop' if 'coop' in a['Kernel_Name'] else 'lane'),dur,gap,a.get('Grid_Size','?')))
print("sum of gaps %.1f us; chain wall %.1f us"%(tot,(int(chain[-1]['End_Timestamp'])-int(chain[0]['Start_Timestamp']))/1e3))
