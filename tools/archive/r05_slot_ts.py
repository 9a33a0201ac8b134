import struct, sys, numpy as np
data=open(sys.argv[1],'rb').read(); off=0
while off < len(data):
    M,Dw,ns,nw=struct.unpack_from("4i",data,off); off+=16
    ts=np.frombuffer(data,dtype=np.uint64,count=nw*128,offset=off).reshape(nw,128).astype(np.int64); off+=nw*1024
    if (Dw&0xffff)!=256 or ns!=14: continue
    ts=ts[ts[:,63]<1000]; nw=len(ts)
    nwv=int(sys.argv[2]) if len(sys.argv)>2 else 4
    start=ts[:,32:32+nwv]; slots=ts[:,64:64+8*nwv].reshape(nw,nwv,8)
    if not slots.all() or not ts[:,32].all(): continue
    d=np.diff(np.concatenate([start[:,:,None],slots],axis=2),axis=2)
    print("stages=%d: per-wave slot durations (clk, median over workgroups), slot 0..7:"%ns)
    for w in range(nwv): print("  wave %d:"%w, " ".join("%5d"%x for x in np.median(d[:,w,:],axis=0)), "  sum %d"%np.median(d[:,w,:].sum(axis=1)))
    st=int(sys.argv[3]) if len(sys.argv)>3 else 5
    print("  tid 0's stage figures: loop %d, epilogue %d; k-loop start of wave 0 is %d clk behind the stage-start stamp, last slot ends %d clk before the loop-end stamp" % (
        np.median(ts[:,2+2*st]-ts[:,1+2*st]), np.median(ts[:,3+2*st]-ts[:,2+2*st]), np.median(ts[:,32]-ts[:,1+2*st]), np.median(ts[:,2+2*st]-ts[:,64+7])))
    break
