import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_shade' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[:12]:
    print(int(r['Grid_Size_X'])//256, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
