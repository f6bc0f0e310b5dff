import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
iters = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
cat = collections.defaultdict(float); cnt = collections.defaultdict(int)
def classify(n):
    for key, name in (("conv_igemm", "igemm"), ("conv3x3_few", "igemm"), ("conv_narrow_in", "igemm"), ("img16_conv3x3", "resident"), ("res8_", "resident"),
                      ("cpool_res", "resident"), ("conv_wgrad", "wgrad"), ("wgrad_reduce_slabs", "wgrad slab reduce / fold"), ("wgrad_cpool_fold", "wgrad slab reduce / fold"),
                      ("cbn_", "cbn"), ("sn_", "sn"), ("prep_", "prep"), ("copyBuffer", "copy"),
                      ("at::native", "torch-native"), ("adam", "adam"), ("pool2x2", "pool"), ("ew_kernel", "elementwise"), ("colsum", "colsum"),
                      ("rng_", "rng"), ("concat", "concat"), ("relu_meanpool", "gap"), ("fillBuffer", "memset")):
        if key in n: return name
    return "other:" + n[:30]
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    c = classify(r['Kernel_Name']); cat[c] += d; cnt[c] += 1
tot = sum(cat.values())
print(f"total {tot/iters:.2f} ms/iter, {sum(cnt.values())/iters:.0f} launches/iter")
for k, v in sorted(cat.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v/iters:7.3f} ms/iter  {100*v/tot:5.1f}%  {cnt[k]/iters:6.1f} launches/iter")
