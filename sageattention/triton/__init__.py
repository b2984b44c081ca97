"""``sageattention.triton`` of the reference: the module names of its Triton quantisers and attention kernels, served by the gfx950 HIP kernels
(there is no Triton in this implementation; the names are kept so that code written against the reference's modules -- its bench scripts --
imports unchanged)."""
