"""Constructor grid shared by make_golden_modules.py (generator) and tests/test_modules_golden.py (data only)."""
CASES = [
    # (dim, ctor kwargs)
    (2, dict(padding='zeros')),
    (2, dict(padding='reflect', init_shift=2, sparsity_term=0.0, active_flag=True)),
    (2, dict(emulate_dw={'kernel_size': 3, 'stride': 1, 'padding': (0, 0)}, init_thumb_rule=2, sparsity_term=0.)),
    (2, dict(emulate_dw={'kernel_size': 3, 'stride': 2, 'padding': 0})),
    (2, dict(emulate_dw={'kernel_size': (5, 3), 'stride': (1, 2), 'padding': (1, 1), 'padding_mode': 'circular'})),
    (2, dict(emulate_dw={'kernel_size': 3, 'stride': 1, 'padding': 1}, padding='symmetric', active_flag=True)),
    (1, dict(emulate_dw={'kernel_size': 5, 'stride': 2, 'padding': 0}, padding='border')),
    (3, dict(emulate_dw={'kernel_size': 3, 'stride': (1, 2, 2), 'padding': (0, 1, 0)}, padding='periodic')),
    (3, dict(padding='zeros', init_shift=(1, 2, 3))),
]
