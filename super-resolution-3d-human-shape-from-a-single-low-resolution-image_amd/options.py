"""Flag system of the reconstruction path.

Same flag names and defaults as the reference's argparse set
(/root/reference/lib/options.py:5-214) so that a reference command line such as
README.md:41-44 parses unchanged.  Flags that only the reference's training or
dataset-rendering code reads are accepted and ignored (listed in `_IGNORED`).
Table-driven instead of one add_argument call per line.
"""
import argparse

# (name, type, default) - flags the hot path reads
_PATH_FLAGS = [
    ("dataroot", str, "./data"),
    ("loadSize", int, 512),
    ("name", str, "example"),
    ("num_views", int, 1),
    ("gpu_id", int, 0),
    ("scale", int, 2),
    ("rgb_range", int, 255),
    ("resolution", int, 512),
    ("z_size", float, 200.0),
    ("norm", str, "group"),
    ("hg_depth", int, 2),
    ("hg_dim", int, 256),
    ("num_stack_lr", int, 3),
    ("num_stack_hr", int, 1),
    ("load_netG_checkpoint_path", str, None),
    ("results_path", str, "./results"),
    ("num_samples", int, 50000),
    ("threshold", float, 0.05),
]
_LIST_FLAGS = [
    ("mlp_dim_lr", int, [321, 1024, 512, 256, 128, 1]),
    ("mlp_dim_hr", int, [322, 1024, 512, 256, 128, 1]),
    ("mlp_res_layers_lr", int, [2, 3, 4]),
    ("mlp_res_layers_hr", int, [2, 3, 4]),
    ("n_block", int, [2, 2, 2]),
    ("b_min", float, [-128.0, -28.0, -128.0]),
    # the reference forgets type=float here (options.py:122); the dataset coerces
    # (EvalDataset_LR_v2.py:145-146).  We coerce at parse time.
    ("b_max", float, [128.0, 228.0, 128.0]),
    ("schedule", int, [60, 80]),
    ("mlp_dim_color", int, [513, 1024, 512, 256, 128, 3]),
]
_BOOL_FLAGS = ["no_residual", "residual", "no_gen_mesh", "debug", "random_multiview", "serial_batches",
               "pin_memory", "no_num_eval", "skip_hourglass", "use_tanh", "random_flip", "random_trans",
               "random_scale", "val_test_error", "val_train_error", "gen_test_mesh", "gen_train_mesh",
               "all_mesh", "with_color", "both_color", "change_weights"]
# accepted for command-line compatibility, never read by this package
_IGNORED = [
    ("gpu_ids", str, "0"), ("num_threads", int, 1), ("batch_size", int, 2), ("learning_rate", float, 1e-3),
    ("learning_rateC", float, 1e-3), ("num_epoch", int, 100), ("freq_plot", int, 10), ("freq_save", int, 50),
    ("freq_save_ply", int, 100), ("resume_epoch", int, -1), ("continue_train", int, -1),
    ("test_folder_path", str, None), ("sigma", float, 5), ("num_sample_inout", int, 6000),
    ("num_sample_color", int, 0), ("norm_color", str, "instance"), ("num_hourglass", int, 2),
    ("hg_down", str, "ave_pool"), ("hourglass_dim", int, 256), ("mlp_norm", str, "group"),
    ("scale_pifu", float, 0.01), ("gamma", float, 0.1), ("color_loss_type", str, "l1"), ("losses", str, "l1"),
    ("mlp1", float, 1.0), ("mlp2", float, 1.0), ("srweight", float, 1.0), ("dispweight", float, 1.0),
    ("disp_error", int, 1), ("n_train", int, 300), ("n_val", int, 60), ("optimizer", str, "ADAM"),
    ("momentum", float, 0.9), ("beta1", float, 0.9), ("beta2", float, 0.999), ("epsilon", float, 1e-8),
    ("ams", float, False), ("weight_decay", float, 0), ("num_gen_mesh_test", int, 1), ("n_colors", int, 3),
    ("checkpoints_path", str, "./checkpoints"), ("load_netC_checkpoint_path", str, None),
    ("load_checkpoint_path", str, None), ("single", str, ""), ("mask_path", str, None), ("img_path", str, None),
    ("aug_alstd", float, 0.0), ("aug_bri", float, 0.0), ("aug_con", float, 0.0), ("aug_sat", float, 0.0),
    ("aug_hue", float, 0.0), ("aug_blur", float, 0.0),
]
# extensions of this package (not in the reference)
_NATIVE = [
    ("precision", str, "fp32"),      # fp32 | bf16 | fp16 : arithmetic of the MLP contractions
    ("encoder_precision", str, "auto"),   # fp32 = auto (two f16 parts, three products per MAC: the parity-grade encoder, with every
                                          # --precision) | f16 (one f16 product per MAC in the 3x3 convolutions: 11 significant bits,
                                          # opt-in; bounded by tests/test_gpu_precision.py with --precision bf16)
    ("encoder_graph", str, "0"),     # 1 = the encoder of a single view replays a captured HIP graph (its outputs are the graph's buffers,
                                     # overwritten by the next call: encoder.py "HIP graphs"; measured 2 % SLOWER than the eager launches
                                     # at 512^2: opt-in) | 0 = eager launches, fresh tensors per call
    ("octree_precision", str, "fp32"),   # fp32 = fp32-grade octree levels whatever --precision says (default) | sweep = the levels in
                                         # --precision's arithmetic (16-bit column kernel: faster, moves the walk's artefact surfaces)
    ("no_octree", None, False),      # dense sweep (the parity target)
    ("synthetic", None, False),      # synthetic image + PRNG weights instead of dataroot / checkpoint
    ("pipeline", None, False),       # eval driver: subjects as a pipeline (train_util.gen_mesh_pipelined) instead of one by one
]


class BaseOptions:
    """Mirror of lib/options.py BaseOptions: initialize / gather_options / parse."""

    def __init__(self):
        self.initialized = False

    def initialize(self, parser):
        for name, typ, default in _PATH_FLAGS + _IGNORED:
            parser.add_argument("--" + name, type=typ, default=default)
        for name, typ, default in _LIST_FLAGS:
            parser.add_argument("--" + name, type=typ, nargs="+", default=list(default))
        for name in _BOOL_FLAGS:
            parser.add_argument("--" + name, action="store_true")
        for name, typ, default in _NATIVE:
            if typ is None:
                parser.add_argument("--" + name, action="store_true")
            elif name == "precision":
                parser.add_argument("--precision", type=str, default=default, choices=["fp32", "bf16", "fp16"])
            elif name == "encoder_precision":
                parser.add_argument("--encoder_precision", type=str, default=default, choices=["auto", "fp32", "f16"])
            else:
                parser.add_argument("--" + name, type=typ, default=default)
        self.initialized = True
        return parser

    def gather_options(self, argv=None):
        parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        if not self.initialized:
            parser = self.initialize(parser)
        self.parser = parser
        return parser.parse_args(argv)

    def print_options(self, opt):
        lines = ["----------------- Options ---------------"]
        for k, v in sorted(vars(opt).items()):
            default = self.parser.get_default(k)
            comment = "" if v == default else "\t[default: %s]" % str(default)
            lines.append("{:>25}: {:<30}{}".format(str(k), str(v), comment))
        lines.append("----------------- End -------------------")
        print("\n".join(lines))

    def parse(self, argv=None):
        return self.gather_options(argv)


def default_opt(**overrides):
    """Options object with the reference defaults (no argv), plus overrides."""
    opt = BaseOptions().parse([])
    for k, v in overrides.items():
        if not hasattr(opt, k):
            raise AttributeError("unknown option " + k)
        setattr(opt, k, v)
    return opt
