#!/usr/bin/env python3
"""Write tests/golden/pyro_store_halfmoons_fc_h32.pt: a Pyro param-store file in pyro-ppl 1.3.0's on-disk layout, built BY HAND
(pyro is not installed in this image, and the reference ships no saved posterior) from the layout of
pyro/params/param_store.py [recalled]:

    ParamStoreDict.save(filename)  ==  torch.save(self.get_state(), f)
    get_state()                    ==  {"params": self._params, "constraints": self._constraints}
      _params[name]       the UNCONSTRAINED value: a leaf torch.Tensor with requires_grad=True
      _constraints[name]  a torch.distributions.constraints object (constraints.real when pyro.param() got none)

with the names the reference's guide registers (model_bnn.py:125-126): "<state_dict key>_loc", "<state_dict key>_scale" for
every key of NN.state_dict() (fc: model.1.weight, model.1.bias, model.3.weight, model.3.bias).  The values are seeded
random numbers; the expected arrays are stored next to the .pt as an .npz so that the loader test needs no pyro either.
PARITY UNPINNED: no pyro-written file exists to compare the layout with.
"""
import os

import numpy as np
import torch
from torch.distributions import constraints

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    shapes = [("model.1.weight", (32, 2)), ("model.1.bias", (32,)), ("model.3.weight", (2, 32)), ("model.3.bias", (2,))]
    g = torch.Generator().manual_seed(20)
    params, expect = {}, {}
    for key, shp in shapes:
        loc = torch.randn(shp, generator=g) * 0.5
        scale = torch.randn(shp, generator=g) * 0.3 - 2.0                    # raw scale: softplus applied at draw time (:127)
        params[key + "_loc"] = loc.clone().requires_grad_(True)
        params[key + "_scale"] = scale.clone().requires_grad_(True)
        expect[key + "_loc"], expect[key + "_scale"] = loc.numpy(), scale.numpy()
    state = {"params": params, "constraints": {name: constraints.real for name in params}}
    torch.save(state, os.path.join(HERE, "pyro_store_halfmoons_fc_h32.pt"))
    np.savez_compressed(os.path.join(HERE, "pyro_store_halfmoons_fc_h32_expected.npz"), **expect)
    print("wrote pyro_store_halfmoons_fc_h32.pt")


if __name__ == "__main__":
    main()
