import sys, time
sys.path[:0]=[".","motion-latent-diffusion_amd"]
import torch, numpy as np
from mld_hip import _lib, synthetic as syn
dev=torch.device("cuda:0")
b = syn.make_batch(64, None, seed=1234, max_len=196); text = torch.from_numpy(b.text_emb).to(dev); m,s=syn.make_mean_std(); w=syn.make_novae_denoiser_state_dict()
e=_lib.Engine(device=0,max_batch=64,max_frames=196,latent_dim=512,vae_arch=_lib.VAE_NONE,denoiser_arch=_lib.ARCH_TRANS_DEC,scheduler_type=_lib.SCHED_DDPM,num_inference_steps=20,steps_offset=0,precision=1)
e.load_state_dict(w,"denoiser."); e.load_tensor("mean",m); e.load_tensor("std",s); e.set_option("range_probe",0); e.finalize()
x0=torch.randn(64,196,263,device=dev); j=torch.empty(64,196,22,3,device=dev)
e.sample_novae(text,x0,b.lengths,None,7,None,j); torch.cuda.synchronize()
