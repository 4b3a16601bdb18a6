IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


class Mixup:  # only referenced as a type annotation on the hot path
    def __init__(self, *a, **k):
        raise NotImplementedError
