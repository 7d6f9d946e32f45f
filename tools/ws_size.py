import sys; sys.path.insert(0,'dl-unet_amd')
import _hip
h=_hip.Handle(64,0)
for B,S,t in ((8,572,1),(8,572,0),(1,572,1),(16,572,1)):
    print(B,S,t, "%.2f GB" % (h.workspace_bytes(B,S,t)/1e9))
