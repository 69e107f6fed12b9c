"""touchnet_amd — MI355X-native (gfx950) packed-sequence multimodal training path.

A from-scratch implementation of ONE hot path of xingchensong/TouchNet: packed
text/audio sequences -> audio frontend -> projector / embedding -> Llama/Qwen2 decoder blocks with
document-masked attention -> per-sentence-normalised cross-entropy, forward and backward, as
hand-written HIP kernels behind TouchNet's own plugin surface (TrainSpec registry, loss/acc functions,
datapipe stage functions).  The HIP extension is mandatory: nothing here falls back to eager PyTorch.
"""
__version__ = "0.1.0"
