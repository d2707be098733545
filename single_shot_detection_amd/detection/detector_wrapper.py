"""DetectorWrapper -- mirror of detection/detector_wrapper.py:27-65 (the TorchScript export path is out of scope)."""
import torch


class DetectorWrapper(object):
    def __init__(self, detector, preprocess, postprocessor):
        self.device = next(detector.parameters()).device
        self.model = detector
        self.preprocess = preprocess
        self.postprocessor = postprocessor
        self.input_size = None
        for transform in getattr(preprocess, 'transforms', []) or []:
            if type(transform).__name__ == 'Resize':
                self.input_size = transform.size
                break

    def eval(self):
        self.model.eval()

    def predict_single(self, img):
        """numpy RGB HxWx3 (or a [3,H,W] tensor when no preprocess is set) -> [K,6] in image coordinates."""
        if self.preprocess is not None:
            ratio_w = img.shape[1] / self.input_size[0]
            ratio_h = img.shape[0] / self.input_size[1]
            img = self.preprocess(img)
        else:
            ratio_w = ratio_h = 1.0
        assert img.dim() == 3
        img = img.unsqueeze(0)
        with torch.no_grad():
            *prediction, priors = self.model(img.to(self.device))
            result = self.postprocessor.postprocess(prediction, priors)[0]
        result[..., [0, 2]] *= ratio_w
        result[..., [1, 3]] *= ratio_h
        return result
