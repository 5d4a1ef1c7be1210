"""Print a rocprofv3 kernel_stats.csv with the kernel-name column cut to 110 characters (every row and every
numeric column kept).  Used to bring the summary back through gpurun's stdout tail."""
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(sys.stdout, quoting=csv.QUOTE_MINIMAL)
for r in rows:
    w.writerow([r[0][:110]] + r[1:])
