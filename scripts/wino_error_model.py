import numpy as np
rng=np.random.RandomState(0)
BT4=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],np.float64)
G4=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],np.float64)
AT4=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],np.float64)
BT2=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64)
G2=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float64)
AT2=np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)
def direct(x,w,dt):
    H,W,C=x.shape; K=w.shape[3]
    xp=np.zeros((H+2,W+2,C),dt); xp[1:-1,1:-1]=x
    y=np.zeros((H,W,K),dt)
    for kh in range(3):
        for kw in range(3):
            y+= (xp[kh:kh+H,kw:kw+W].reshape(-1,C).astype(dt)@w[kh,kw].astype(dt)).reshape(H,W,K)
    return y
def wino(x,w,BT,G,AT,m,dt):
    H,W,C=x.shape; K=w.shape[3]; a=m+2
    TY,TX=-(-H//m),-(-W//m)
    xp=np.zeros((TY*m+2,TX*m+2,C),dt); xp[1:H+1,1:W+1]=x
    U=np.einsum('ik,klcn,jl->ijcn',G,w.astype(np.float64),G).astype(dt)   # weights transformed in f64 then rounded (kernel does fp32; close)
    BTd=BT.astype(dt); ATd=AT.astype(dt)
    # tiles
    d=np.zeros((TY,TX,a,a,C),dt)
    for i in range(a):
        for j in range(a):
            d[:,:,i,j]=xp[i:i+TY*m:m, j:j+TX*m:m][:TY,:TX]
    # V = BT d B  in dt
    t=np.zeros_like(d)
    for xi in range(a):
        acc=np.zeros((TY,TX,a,C),dt)
        for i in range(a):
            if BT[xi,i]!=0: acc=acc+BTd[xi,i]*d[:,:,i]
        t[:,:,xi]=acc
    V=np.zeros_like(d)
    for nu in range(a):
        acc=np.zeros((TY,TX,a,C),dt)
        for j in range(a):
            if BT[nu,j]!=0: acc=acc+BTd[nu,j]*t[:,:,:,j]
        V[:,:,:,nu]=acc
    M=np.zeros((TY,TX,a,a,K),dt)
    for xi in range(a):
        for nu in range(a):
            M[:,:,xi,nu]=(V[:,:,xi,nu].reshape(-1,C)@U[xi,nu]).reshape(TY,TX,K)
    # Y = AT M A
    z=np.zeros((TY,TX,m,a,K),dt)
    for yy in range(m):
        acc=np.zeros((TY,TX,a,K),dt)
        for xi in range(a):
            if AT[yy,xi]!=0: acc=acc+ATd[yy,xi]*M[:,:,xi]
        z[:,:,yy]=acc
    Y=np.zeros((TY,TX,m,m,K),dt)
    for xx in range(m):
        acc=np.zeros((TY,TX,m,K),dt)
        for nu in range(a):
            if AT[xx,nu]!=0: acc=acc+ATd[xx,nu]*z[:,:,:,nu]
        Y[:,:,:,xx]=acc
    return Y.transpose(0,2,1,3,4).reshape(TY*m,TX*m,K)[:H,:W]
for (H,W,C,K) in [(28,28,256,64),(28,28,512,64),(56,56,128,64),(32,24,512,64)]:
    x=np.maximum(rng.randn(H,W,C),0).astype(np.float32)
    w=(rng.randn(3,3,C,K)*np.sqrt(2.0/(9*C))).astype(np.float32)
    ref=direct(x,w,np.float64)
    rngv=np.abs(ref).max()
    for name,y in [('direct32',direct(x,w,np.float32)),('F2x2',wino(x,w,BT2,G2,AT2,2,np.float32)),('F4x4',wino(x,w,BT4,G4,AT4,4,np.float32)),('F4x4_f64',wino(x,w,BT4,G4,AT4,4,np.float64))]:
        e=np.abs(y-ref)
        print(H,W,C,K,name,'max err/range %.2e  rms err/rms %.2e'%(e.max()/rngv, np.sqrt((e**2).mean())/np.sqrt((ref**2).mean())))
