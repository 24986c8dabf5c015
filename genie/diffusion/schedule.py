from genie2_amd.pack import cosine_betas as cosine_beta_schedule  # noqa: F401


def get_betas(n_timestep, schedule):
    if schedule != 'cosine':
        raise ValueError('Invalid schedule: {}'.format(schedule))
    return cosine_beta_schedule(n_timestep)
