"""Worker of tests/test_gpu_distributed.py: one rank of a P-rank run of the Component /
interactions.gravity() layer over x-slab domains.  The ranks share cuda:0 and talk over gloo
(test only; production is one GPU per rank over RCCL).  Every rank runs the SAME test body
the single-domain suite runs — gravity(), Component.drift(), stepper.timeloop, RungStepper are
collective and unchanged — and asserts on the gathered results."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def golden(name):
    return np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz'))


def main():
    case, arg = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from concept_amd import comm
    comm.init()
    assert comm.active() is not None and comm.active().world == world
    if case == 'steps':
        import test_gpu_p3m
        test_gpu_p3m.test_timeloop_sequence_vs_reference(golden, arg)
    elif case == 'rungs':
        import test_gpu_p3m
        test_gpu_p3m.test_adaptive_rungs_vs_reference(golden, None, None)
        test_gpu_p3m.test_adaptive_rungs_knot_across_domains()
    elif case == 'p3m_kick':
        import test_gpu_p3m
        test_gpu_p3m.test_shortrange_vs_golden_and_oracle(golden, arg, None, None)
        test_gpu_p3m.test_p3m_full_kick_any(golden)
    elif case == 'mixed':
        import test_gpu_fluid
        test_gpu_fluid.test_mixed_pm_vs_golden(golden, arg)
    elif case == 'nonlinnu':
        import test_gpu_fluid
        test_gpu_fluid.test_nonlinnu_shape_three_interactions(golden)
    elif case == 'multigrid':
        import test_gpu_fluid
        test_gpu_fluid.test_multigrid_vs_golden(golden, arg)
    elif case == 'orders':
        import test_gpu_fluid
        test_gpu_fluid.test_orders_interlacing_fourier_diff_vs_golden(golden, arg)
    elif case == 'tiled_general':
        import test_gpu_fluid
        test_gpu_fluid.test_general_path_uses_tile_order(golden)
    elif case == 'diff_orders':
        import test_gpu_pm
        test_gpu_pm.test_other_differentiation_orders_vs_golden(torch, golden, arg)
    elif case == 'pp':
        import test_gpu_pp
        name, method = arg.split(',')
        test_gpu_pp.test_pp_vs_golden_and_oracle(golden, name, method)
        if method == 'pp':
            test_gpu_pp.test_pp_default_ewald_grid_and_larger_set()
    elif case == 'known':
        import test_gpu_known_answers as k
        {'k2': k.test_k2_six_symmetric_particles_fall_to_the_centre,
         'k3': k.test_k3_two_groups_of_four_keep_identical_x}[arg]()
    elif case == 'mixed_random':
        import test_gpu_fluid
        test_gpu_fluid.test_mixed_pm_vs_oracle_random()
    elif case == 'random_configs':
        import test_gpu_fluid
        import test_gpu_p3m
        for seed in range(int(arg)):
            test_gpu_fluid.test_random_configurations_vs_oracle(seed)
        import test_gpu_pm
        for seed in range(8):
            test_gpu_pm.test_random_streaming_timeloops(torch, seed)
        for seed in range(8):
            test_gpu_p3m.test_random_p3m_timeloops_across_domains(seed)
        for seed in range(int(arg)):
            try:
                test_gpu_p3m.test_random_shortrange_vs_oracle(seed)
            except Exception as e:  # a slab thinner than the force range is refused, not wrong
                if 'too large for slabs of width' not in str(e):
                    raise
    elif case == 'advice':
        import test_gpu_p3m
        test_gpu_p3m.test_shortrange_two_components_receivers_not_suppliers(arg == 'cell')
        test_gpu_p3m.test_populate_after_sort_lands_on_the_right_particles()
    elif case == 'traj':
        import test_gpu_trajectory as tt
        tt.test_timeloop_run_vs_reference(golden, arg)
        if 'traj_pm' in arg:
            for streaming in (True, False):
                tt.test_stepper_timeloop_replays_reference_integrals(golden, arg, streaming)
    elif case == 'config4':
        import test_gpu_fluid
        n_side, gs = (int(v) for v in arg.split(','))
        test_gpu_fluid.test_config4_shape_across_domains(n_side, gs)
    elif case == 'snapshot':
        import test_gpu_pp
        test_gpu_pp.test_gadget_snapshot_to_gpu_components(golden)
    elif case == 'void':
        import test_gpu_pm
        from concept_amd import stepper
        test_gpu_pm.test_void_domains_vs_oracle(torch, False)
        replays = stepper.stream_replays
        test_gpu_pm.test_void_domains_vs_oracle(torch, True)
        assert stepper.stream_replays > replays  # (the row buffer really overflowed)
        test_gpu_pm.test_void_domains_vs_oracle(torch, 'point')
        test_gpu_pm.test_void_domains_p3m(torch, False)
        test_gpu_pm.test_void_domains_p3m(torch, True)
    elif case == 'pm_api':
        import test_gpu_pm
        test_gpu_pm.test_gravity_api_pm(None, golden)
    elif case == 'k1':
        import test_gpu_known_answers
        test_gpu_known_answers.k1_lattice(int(arg), n_lin=16, gridsize=32)
    elif case == 'k4':
        import test_gpu_known_answers
        gs, order = arg.split(',')
        test_gpu_known_answers.test_k4_sine_wave_fluid_stays_uniform_in_yz(int(gs), int(order))
    else:
        raise SystemExit(f'unknown case {case}')
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(f'RANK{rank}-OK')


if __name__ == '__main__':
    main()
