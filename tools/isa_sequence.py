import re,sys,subprocess
src=sys.argv[1]; pat=sys.argv[2]; minm=int(sys.argv[3]) if len(sys.argv)>3 else 20
extra=sys.argv[4:] 
subprocess.run(['/opt/rocm/bin/hipcc','--offload-arch=gfx950','-O3','-std=c++17','-fPIC','-munsafe-fp-atomics','-I/root/repo/include','-I/root/repo/repmode_amd/csrc','-S','--cuda-device-only',src,'-o','/tmp/x.s']+extra,stderr=subprocess.DEVNULL)
s=open('/tmp/x.s').read()
for m in re.finditer(r'^(_Z\S+):\s*; @', s, re.M):
    n=m.group(1)
    dem=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    dem=dem.replace('(anonymous namespace)::','')
    if not re.search(pat, dem): continue
    start=m.start(); end=s.index('.Lfunc_end', start)
    body=s[start:end]
    vg=re.search(re.escape(n)+r'\.num_vgpr, (\d+)', s); 
    sp=re.search(r'; ScratchSize: (\d+)', s[end:end+3000])
    print('==',dem[:140],'vgpr',vg.group(1) if vg else '?','scratch',sp.group(1) if sp else '?')
    blocks=[]; cur=['entry',[]]
    for l in body.split('\n'):
        mm=re.match(r'^(\.LBB[0-9_]+):',l)
        if mm: blocks.append(cur); cur=[mm.group(1),[]]
        else: cur[1].append(l.strip())
    blocks.append(cur)
    for name,ls in blocks:
        nm=sum(1 for l in ls if l.startswith('v_mfma'))
        if nm<minm: continue
        seq=[]
        for l in ls:
            if not l or l.startswith(';'): continue
            op=l.split()[0]
            if op.startswith('v_mfma'): seq.append('M')
            elif op.startswith('ds_read'): seq.append('d')
            elif op.startswith('ds_write'): seq.append('D')
            elif op.startswith('global_load') or op.startswith('buffer_load'): seq.append('G')
            elif op.startswith('global_store') or op.startswith('buffer_store') or op.startswith('global_atomic'): seq.append('S')
            elif op=='s_waitcnt': seq.append('w['+l.split(None,1)[1].replace(' ','').replace('vmcnt','v').replace('lgkmcnt','l')+']')
            elif op=='s_barrier': seq.append('BAR')
            elif op.startswith('v_'): seq.append('v')
            elif op.startswith('s_'): seq.append('s')
            else: seq.append('?')
        print(name,'mfma',nm); print(' '.join(seq))
