import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for p in f:
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'][:60]; agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        if r['Counter_Name']=='SQ_WAVES': cnt[k]+=1
for k,v in agg.items():
    w=v.get('SQ_WAVES',1) or 1
    print(k, 'launches',cnt[k], {c: round(x/w,1) for c,x in v.items()})
