import math
import numpy as np
from scipy.linalg import expm
def hat3(w): return np.array([[0,-w[2],w[1]],[w[2],0,-w[0]],[-w[1],w[0],0.]])
def hat(xi):   # xi = (omega, p) as in light.hip hat_into
    H=np.zeros((4,4)); H[:3,:3]=hat3(xi[:3]); H[:3,3]=xi[3:]; return H
def coeffs(th2):
    # A=sin/th, B=(1-cos)/th^2, C=(th-sin)/th^3, D=(th^2+2cos-2)/(2 th^4), E=(2th-3sin+th cos)/(2 th^5)
    if th2 < 0.25:
        # series in x = th^2
        x=th2
        A=1-x/6*(1-x/20*(1-x/42*(1-x/72*(1-x/110*(1-x/156)))))
        B=0.5*(1-x/12*(1-x/30*(1-x/56*(1-x/90*(1-x/132*(1-x/182))))))
        C=(1/6)*(1-x/20*(1-x/42*(1-x/72*(1-x/110*(1-x/156*(1-x/210))))))
        # D = sum_{k>=0} (-1)^k x^k * 2/(2k+4)! *... derive: cos = sum (-1)^n th^(2n)/(2n)!; th^2+2cos-2 = 2*sum_{n>=2} (-1)^n th^(2n)/(2n)!; /(2 th^4) = sum_{n>=2} (-1)^n x^(n-2)/(2n)!
        D=sum((-1)**n * x**(n-2)/math.factorial(2*n) for n in range(2,10))
        # E: 2th-3sin+th cos = sum_n (-1)^n th^(2n+1) [ -3/(2n+1)! + 1/(2n)! ] for n>=1 (n=0: 2-3+1=0) ; n=1: -(-3/6+1/2)=0 ; so starts n=2
        E=sum((-1)**n * x**(n-2) * (1/math.factorial(2*n) - 3/math.factorial(2*n+1)) for n in range(2,10))/2
        return A,B,C,D,E
    th=np.sqrt(th2); s,c=np.sin(th),np.cos(th)
    return s/th,(1-c)/th2,(th-s)/(th2*th),(th2+2*c-2)/(2*th2*th2),(2*th-3*s+th*c)/(2*th2*th2*th)
def closed(xi):
    w,p=xi[:3],xi[3:]; th2=w@w; A,B,C,D,E=coeffs(th2)
    W=hat3(w); P=hat3(p); W2=W@W
    R=np.eye(3)+A*W+B*W2; J=np.eye(3)+B*W+C*W2; t=J@p
    Q=0.5*P + C*(W@P+P@W+W@P@W) + D*(W2@P+P@W2-3*W@P@W) + E*(W@P@W2+W2@P@W)
    # twists v_i=(omega_i,u_i): i<3 rotation generator: omega=J e_i, u=Q e_i ; i>=3: omega=0,u=J e_{i-3}
    V=np.zeros((6,6))
    for i in range(3): V[i,:3]=J[:,i]; V[i,3:]=Q[:,i]
    for i in range(3): V[3+i,3:]=J[:,i]
    return R,t,V
def numeric(xi):
    T=expm(hat(xi)); V=np.zeros((6,6))
    for i in range(6):
        e=np.zeros(6); e[i]=1
        S=np.zeros((8,8)); S[:4,:4]=hat(xi); S[4:,4:]=hat(xi); S[:4,4:]=hat(e)
        Dm=expm(S)[:4,4:]
        Hm=Dm@np.linalg.inv(T)
        V[i,:3]=[Hm[2,1],Hm[0,2],Hm[1,0]]; V[i,3:]=Hm[:3,3]
    return T[:3,:3],T[:3,3],V
rng=np.random.default_rng(0)
worst=0
for scale in (0,1e-6,1e-3,0.05,0.3,0.49,0.51,1.0,2.5,3.1):
    for _ in range(20):
        xi=rng.normal(size=6); xi[:3]*=scale/max(np.linalg.norm(xi[:3]),1e-300) if scale>0 else 0; xi[3:]*=rng.uniform(0,2)
        R,t,V=closed(xi); Rn,tn,Vn=numeric(xi)
        e=max(abs(R-Rn).max(),abs(t-tn).max(),abs(V-Vn).max()); worst=max(worst,e)
    print(scale, 'max err so far', worst)
