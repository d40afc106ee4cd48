import numpy as np, torch, sys, time
sys.path.insert(0,'.')
import spectrogram_amd as S, oracle as O
from tests.conftest import power_error
for n,hop in [(1024,256),(256,128),(2048,512),(4096,1024),(1024,100),(32,8),(16384,4096)]:
    rng=np.random.default_rng(1)
    x=rng.uniform(-1,1,(2,n+50*hop)).astype(np.float32)
    with S.Engine(fft_size=n,hop=hop) as e:
        got=e.stft(torch.from_numpy(x).cuda()); torch.cuda.synchronize(); got=got.cpu().numpy()
    ref,_=O.stft(O.OracleConfig.make(n,hop),x[1])
    print(n,hop,'err',power_error(got[1],ref), 'maxabs/peak', np.abs(got[1]-ref).max()/ref.max(), flush=True)
