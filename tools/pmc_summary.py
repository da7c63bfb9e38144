import csv, collections, glob, sys
def load(d):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ","").replace("cl2::","")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == list(agg[k].keys())[0]: n[k]+=1
    return agg, n
for d in sys.argv[1:]:
    agg, n = load(d)
    for k in sorted(agg, key=lambda k:-sum(agg[k].values())):
        if "rocclr" in k or "export" in k: continue
        print(f"{k[:44]:44s} n={n[k]:3d} " + " ".join(f"{c.replace('SQ_','')}={v/n[k]:.3g}" for c,v in agg[k].items()))
