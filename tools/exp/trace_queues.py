#!/usr/bin/env python3
import csv, glob, sys
from collections import Counter
c = Counter()
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        k = 'fit' if 'fit_grad' in n else 'match' if 'match_kernel' in n else 'copy' if 'copyBuffer' in n else 'sucre-other' if 'sucre::' in n else 'torch'
        c[(k, r['Queue_Id'], r['Stream_Id'], r['Thread_Id'])] += 1
for k, v in sorted(c.items()):
    print(k, v)
