import re,sys
def funcs(path, ren):
    s=open(path).read()
    s=re.sub(r'__hip_cuid_[0-9a-f]+','__hip_cuid_X',s)
    for a,b in ren: s=s.replace(a,b)
    d={}
    for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)^\.Lfunc_end\d+:', s, re.S|re.M):
        body=re.sub(r'\.L(BB|tmp|func_begin|func_end)\d+(_\d+)?','.L',m.group(2))
        body=re.sub(r';.*','',body)
        d[m.group(1)]=body
    return d
ren=[('k_attention_fwdILb0ELb0ELb0ELb0EE','k_attention_fwdILb0ELb0ELb0EE'),('k_attention_fwdILb0ELb0ELb1ELb1EE','k_attention_fwdILb0ELb1ELb1EE'),
     ('k_attention_fwdILb1ELb0ELb0ELb0EE','k_attention_fwdILb1ELb0ELb0EE'),('k_attention_fwdILb1ELb0ELb0ELb1EE','k_attention_fwdILb1ELb0ELb1EE')]
for f in sys.argv[3:]:
    a=funcs('%s/%s.s'%(sys.argv[1],f),ren); b=funcs('%s/%s.s'%(sys.argv[2],f),ren)
    print(f,len(a),len(b))
    for k in a:
        if k not in b: print('  removed',k)
        elif a[k]!=b[k]: print('  DIFF',k,len(a[k]),len(b[k]))
    for k in b:
        if k not in a: print('  added',k)
