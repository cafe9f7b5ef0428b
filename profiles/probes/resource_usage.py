import re,subprocess,sys
txt=open(sys.argv[1]).read()
pat=sys.argv[2] if len(sys.argv)>2 else ''
blocks=re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
rows=[]
for b in blocks:
    name=b.split('\n')[0].split()[0]
    def g(k):
        m=re.search(k+r': (\d+)', b); return int(m.group(1)) if m else -1
    rows.append((name,g('VGPRs'),g('AGPRs'),g(r'ScratchSize \[bytes/lane\]'),g(r'Occupancy \[waves/SIMD\]'),g(r'LDS Size \[bytes/block\]')))
dem=subprocess.run(['c++filt']+[r[0] for r in rows],capture_output=True,text=True).stdout.splitlines()
for d,r in zip(dem,rows):
    if re.search(pat,d): print('vgpr %3d agpr %3d scratch %3d occ %d lds %6d  %s'%(r[1],r[2],r[3],r[4],r[5],d[:130]))
