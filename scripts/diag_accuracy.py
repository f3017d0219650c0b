import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import oracle_lib as ol, __graft_entry__ as ge
pkg=ge.load(); fa=pkg.filterapi
def run(L,M,fs,nch):
    N=L+M-1
    kinds=[(50/12000,3000/12000),(-200/12000,200/12000),(-5000/12000,5000/12000)]
    plan=[]
    for i in range(nch):
        f=1e6+i*60e3+(i%40) if 1e6+nch*60e3<fs/2 else 0.02*fs+i*(0.46*fs/nch)+(i%40)
        plan.append((ol.compute_tuning(N,fs,f)[1],)+kinds[i%3])
    master=fa.create_filter_input(L,M,fa.REAL)
    slaves=[]
    for sh,lo,hi in plan:
        s=fa.create_filter_output(master,240,fa.COMPLEX); fa.set_filter(s,lo,hi,11.0); slaves.append(s)
    gen=ol.SigGen(10.00002e6/fs,0.1,0.01,ol.scale_ad(True,1),True,seed=1)
    st=ol.Stream(L,M,ol.REAL)
    R=ol.ref()
    res={}
    x=[gen.generate(L) for _ in range(2)]
    for prec in (0,1):
        R.oracle_fft_set_precision(prec)
        rm=ol.RefMaster(L,M,ol.REAL); rc=[]
        for sh,lo,hi in plan[:nch]:
            c=rm.channel(240,ol.COMPLEX); c.set_filter(lo,hi,11.0); rc.append(c)
        outs=[]
        for b in range(2):
            rm.write(x[b]); outs.append([c.execute(p[0]) for c,p in zip(rc,plan)])
        res[prec]=outs; rm.close()
    R.oracle_fft_set_precision(0)
    eg=[];e0=[];e1=[]; rmsl=[]; mar=[]
    for b in range(2):
        fa.write_rfilter(master,x[b]); spec64=st.push(x[b],f64=True)
        gs=master._engine.spectrum(b%4)
        print('spec rel err', np.linalg.norm(gs-spec64)/np.linalg.norm(spec64), 'spec rms', np.sqrt(np.mean(abs(spec64)**2)), 'abs err rms', np.sqrt(np.mean(abs(gs-spec64)**2)))
        for i,(s,p) in enumerate(zip(slaves,plan)):
            fa.execute_filter_output(s,p[0])
            want=ol.channel(spec64,ol.REAL,300,240,p[0],s.response)
            nr=np.linalg.norm(want); rmsl.append(nr/np.sqrt(240))
            fl=2e-8*np.abs(spec64).max()*np.linalg.norm(s.response); mar.append((np.linalg.norm(s.output-want)/np.sqrt(240))/(1e-5*nr/np.sqrt(240)+fl))
            eg.append(np.linalg.norm(s.output-want)/nr); e0.append(np.linalg.norm(res[0][b][i]-want)/nr); e1.append(np.linalg.norm(res[1][b][i]-want)/nr)
    eg,e0,e1,rmsl=map(np.array,(eg,e0,e1,rmsl))
    for name,e in (('gpu',eg),('ref f64fft',e0),('ref f32fft',e1)):
        print(name,'median %.2e  p90 %.2e  max %.2e'%(np.median(e),np.percentile(e,90),e.max()), 'argmax',e.argmax(), 'kind',e.argmax()%nch%3, 'rms',rmsl[e.argmax()])
    print('abs err rms gpu (median/max):', np.median(eg*rmsl), (eg*rmsl).max(), ' f32ref:', np.median(e1*rmsl),(e1*rmsl).max(), 'chan rms min/median/max', rmsl.min(), np.median(rmsl), rmsl.max())
    print('tolerance usage (err / allowed): median %.3f max %.3f'%(np.median(mar),max(mar)))
    fa.delete_filter_input(master)
run(1296000,324001,64.8e6,256)
run(2592000,648001,129.6e6,256)
