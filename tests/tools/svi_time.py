import torch, time, sys
sys.path.insert(0, ".")
from oracle import bnn_oracle as O
from robustbnns_amd.model_bnn import BNN
from robustbnns_amd import adversarialAttacks as A
C,H,S,N = 10,512,100,10000
x,y = O.synthetic_inputs(N,(1,28,28),C,seed=2)
xd = x.cuda(); lab = y.argmax(-1).cuda()
g = torch.Generator().manual_seed(3)
for inf in ("hmc","svi"):
    bnn = BNN("mnist", H, "leaky", "fc", inf, 1, 0.01, S, 0, (1,28,28), C)
    if inf == "svi":
        loc = {k: torch.randn(v.shape, generator=g)*0.05 for k,v in bnn.basenet.state_dict().items()}
        scale = {k: torch.full(v.shape, -3.0) for k,v in bnn.basenet.state_dict().items()}
        bnn.set_variational_params(loc, scale, "cuda:0")
    else:
        post = O.synthetic_posterior("fc", 784, H, C, S, 0.05)
        bnn.set_posterior_samples(post, "cuda:0")
    for name, fn in (("forward", lambda: bnn.forward(xd, n_samples=S)), ("fgsm_attack(all points)", lambda: A.fgsm_attack(bnn, xd, lab, {"epsilon":0.3}, n_samples=S))):
        fn(); torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
        print(inf, name, "%.2f ms" % (dt*1e3))
    t0=time.perf_counter(); A.pgd_attack(bnn, xd, lab, {"epsilon":0.3}, n_samples=S); torch.cuda.synchronize(); print(inf, "pgd_attack T=40 (all points): %.1f ms" % ((time.perf_counter()-t0)*1e3))
    onehot = torch.nn.functional.one_hot(lab, C).float()
    t0=time.perf_counter(); r = A.attack_evaluation(bnn, xd, xd, onehot, "cuda:0", n_samples=S); torch.cuda.synchronize(); print(inf, "attack_evaluation: %.1f ms" % ((time.perf_counter()-t0)*1e3))
    from robustbnns_amd import lossGradients as LG
    t0=time.perf_counter(); g_ = LG.loss_gradient(bnn, xd[0], onehot[0], n_samples=S); torch.cuda.synchronize(); print(inf, "loss_gradient (one point): %.2f ms" % ((time.perf_counter()-t0)*1e3))
