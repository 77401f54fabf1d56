"""Command-line flags of the hot-path scripts.  Names and defaults are the reference's
(src/arguments.py:3-68, plus behavioral_cloning/save_embedded_obs.py:25-26); the only additions are the
MI355X knobs at the end.  A fresh parser is built per call (the reference mutates one module-level parser,
which makes a second import of a script fail with an argparse conflict, cf. main_test.py:14)."""
import argparse


def make_parser():
    parser = argparse.ArgumentParser(description='PyTorch Scalable Agent')
    # Behavioral Cloning Settings.
    parser.add_argument('--max_frames', type=int, default=200000000)
    parser.add_argument('--n_episodes_test', type=int, default=50)
    parser.add_argument('--eval_frequency', type=int, default=200)
    parser.add_argument('--to_env', type=str, default='HabitatImageNav-apartment_0')
    parser.add_argument('--debug', action='store_true')
    parser.add_argument('--disable_save', action='store_true')
    parser.add_argument('--essential_save_only', action='store_true')
    parser.add_argument('--save_path', type=str, default='bc')
    parser.add_argument('--data_path', type=str, default='behavioral_cloning')
    # Embedding Settings.
    parser.add_argument('--embedding_name', type=str, default='resnet50')
    parser.add_argument('--train_embedding', action='store_true')
    parser.add_argument('--disable_pretrained_embedding', action='store_false', dest='pretrained_embedding')
    parser.add_argument('--batch_norm', action='store_true')
    # not in the reference (src/arguments.py): 5 = corner + centre windows per frame (BASELINE config 5 extension), 1 = CenterCrop
    parser.add_argument('--crops', type=int, default=1, choices=[1, 5])
    # Environment Settings.
    parser.add_argument('--env', type=str, default='HabitatImageNav-apartment_0')
    parser.add_argument('--num_input_frames', type=int, default=1)
    # General Settings.
    parser.add_argument('--xpid', default=None)
    parser.add_argument('--run_id', default=1, type=int)
    parser.add_argument('--seed', default=1, type=int)
    # Training settings.
    parser.add_argument('--total_frames', default=50000000, type=int)
    parser.add_argument('--batch_size', default=32, type=int)
    parser.add_argument('--unroll_length', default=100, type=int)
    parser.add_argument('--mp_start', default='spawn', type=str)
    parser.add_argument('--disable_cuda', action='store_true')
    # Optimizer settings.
    parser.add_argument('--learning_rate', default=0.0001, type=float)
    parser.add_argument('--alpha', default=0.99, type=float)
    parser.add_argument('--momentum', default=0, type=float)
    parser.add_argument('--epsilon', default=1e-5, type=float)
    parser.add_argument('--max_grad_norm', default=40., type=float)
    # save_embedded_obs.py:25-26
    parser.add_argument('--n_trajectories', type=int, default=-1)
    parser.add_argument('--source', type=str, default='png', choices=['png', 'pickle'])
    # MI355X additions (not in the reference)
    parser.add_argument('--compute_dtype', type=str, default=None, choices=[None, 'bf16', 'f16'],
                        help='encoder storage/MFMA input type (default: $PVR_DTYPE or f16 = inside the 1e-3 parity bound; bf16 = wider range, 3e-3)')
    parser.add_argument('--embed_batch', type=int, default=256, help='frames per encoder launch (the reference '
                        'pushes batch_size x n_frames = 64 per forward, save_embedded_obs.py:151-153)')
    parser.add_argument('--embed_block', type=int, default=0, help='observation rows a rank reads, embeds and appends to its shard file at a '
                        'time (bounds host memory for scenes of millions of frames); 0 = about 1 GiB of frames')
    parser.add_argument('--num_actions', type=int, default=3, help='size of the policy head when no simulator is attached (the reference '
                        'reads env.gym_env.action_space.n, main_bc_2.py:77; Habitat ImageNav without STOP has 3 actions, gym_wrappers.py:173). '
                        'The data is checked against it - it is never derived from the data')
    parser.add_argument('--autograd_step', action='store_true', help="run the reference's own training lines (loss.backward(), "
                        'clip_grad_norm_, torch.optim.RMSprop.step(): main_bc_2.py:206-227) through the autograd bridge instead of the fused step')
    parser.add_argument('--optimizer', type=str, default='rmsprop', choices=['rmsprop', 'adam'],
                        help="'rmsprop' is the reference's optimiser (main_bc_2.py:80-86); 'adam' is an extension")
    return parser


parser = make_parser()
