import types


class Experimenter:
    def __init__(self, model_dir, cfg_path):
        self.model_dir, self.cfg_path = model_dir, cfg_path
        self.config = types.SimpleNamespace(RPN3D_ENABLE=True, min_depth=2.0, max_depth=40.4, loss_disp=True, PlaneSweepVolume=True,
                                            debug=False, valid_classes=[2], eval_depth=True, learn_viewpoint=False)


def mem_info():
    return [10, 3]
