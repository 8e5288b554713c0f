import torch, richdem_amd as rd
n=40000
Z=torch.empty((n,n),dtype=torch.float32,device="cuda"); rd.synth_dem_dev(Z,seed=3); rd.fill_depressions_dev(Z)
d=torch.empty((n,n),dtype=torch.uint8,device="cuda")
rd.d8_flow_directions_dev(Z,-9999.0,d,flats=True); torch.cuda.synchronize()
