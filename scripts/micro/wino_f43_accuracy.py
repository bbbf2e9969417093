"""Accuracy of a 3-tap correlation along x over K = 64 x 9 fp32-accumulated terms with split-fp16 three-pass products:
direct, Winograd F(2,3) (what conv3d_wino.hip does) and F(4,3) (HISTORY.md section 7, item 1a), against float64.
  python scripts/micro/wino_f43_accuracy.py   ->   direct 3.9e-07, F(2,3) 8.3e-07, F(4,3) 7.3e-06 (max, relative to max|y|)"""
import numpy as np
rng=np.random.default_rng(0)
K=64*9        # cin * (kd,kh) terms summed in fp32
X=66; nout=64
d=rng.standard_normal((K,X)).astype(np.float32)          # inputs after GN affine ~N(0,1)
g=(rng.standard_normal((K,3))*0.05).astype(np.float32)
ref=np.zeros(nout); 
for i in range(nout):
    ref[i]=np.sum(d[:,i:i+3].astype(np.float64)*g.astype(np.float64))
def split16(x):
    hi=x.astype(np.float16).astype(np.float32)   # RNE; (pkrtz truncation similar)
    lo=(x-hi).astype(np.float16).astype(np.float32)
    return hi,lo
def prod3(a,b):
    ah,al=split16(a); bh,bl=split16(b)
    return (ah*bh+ (ah*bl + al*bh))   # fp32 accumulate approx
# direct with split products, fp32 accumulation
def direct():
    y=np.zeros(nout,np.float32)
    for i in range(nout):
        acc=np.float32(0)
        p=prod3(d[:,i:i+3],g)          # K x 3
        acc=np.sum(p.astype(np.float32),dtype=np.float32)
        y[i]=acc
    return y
def f23():
    y=np.zeros(nout,np.float32)
    U=np.stack([g[:,0],(g[:,0]+g[:,1]+g[:,2])*np.float32(0.5),(g[:,0]-g[:,1]+g[:,2])*np.float32(0.5),g[:,2]],1).astype(np.float32)
    for i in range(0,nout,2):
        dd=d[:,i:i+4]
        V=np.stack([dd[:,0]-dd[:,2],dd[:,1]+dd[:,2],dd[:,2]-dd[:,1],dd[:,1]-dd[:,3]],1).astype(np.float32)
        m=np.sum(prod3(V,U),0,dtype=np.float32)
        y[i]=m[0]+m[1]+m[2]; y[i+1]=m[1]-m[2]-m[3]
    return y
def f43():
    y=np.zeros(nout,np.float32)
    f=np.float32
    G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],np.float64)
    U=(g.astype(np.float64)@G.T).astype(np.float32)      # weights transformed offline in fp64 -> fp32
    BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],np.float32)
    AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],np.float32)
    for i in range(0,nout,4):
        dd=d[:,i:i+6]
        V=(dd@BT.T).astype(np.float32)
        m=np.sum(prod3(V,U),0,dtype=np.float32)
        y[i:i+4]=(AT@m).astype(np.float32)
    return y
for name,fn in (("direct",direct),("F(2,3)",f23),("F(4,3)",f43)):
    y=fn(); e=np.abs(y-ref).max()/np.abs(ref).max(); print(name,"max rel err %.2e"%e, "rms %.2e"%(np.sqrt(np.mean((y-ref)**2))/np.sqrt(np.mean(ref**2))))
