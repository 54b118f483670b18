"""Summarise the counter CSVs written by tools/micro/pmc_conv.sh: per-dispatch means for the conv kernel."""
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/p*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'conv_mfma_kernel' in r['Kernel_Name'] or 'conv3x3_pipe' in r['Kernel_Name'] or 'stream' in r['Kernel_Name'] or 'head_cls_rows' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = acc[k]
    print('%-34s %14.0f  (n=%d)' % (k, sum(v) / len(v), len(v)))
