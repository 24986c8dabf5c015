"""`Config`: the reference's whitespace `key value` configuration file
(genie/config.py:9-107) -> io / diffusion / model / training / optimization
dictionaries with the same keys and defaults."""

_SPEC = {
    'io': [
        ('name', 'name', None, str), ('rootdir', 'rootDirectory', 'runs', str),
        ('datadir', 'dataDirectory', 'data/afdbreps_l-256_plddt_80/pdbs', str),
        ('min_n_res', 'minimumNumResidues', 20, int), ('max_n_res', 'maximumNumResidues', 256, int),
        ('max_n_chain', 'maximumNumChains', 1, int), ('validation_split', 'validationSplit', None, float),
        ('motif_prob', 'motifProbability', 0.8, float), ('motif_min_pct_res', 'motifMinimumPercentageResidues', 0.05, float),
        ('motif_max_pct_res', 'motifMaximumPercentageResidues', 0.5, float),
        ('motif_min_n_seg', 'motifMinimumNumberSegments', 1, int), ('motif_max_n_seg', 'motifMaximumNumberSegments', 4, int),
    ],
    'diffusion': [('n_timestep', 'numTimesteps', 1000, int), ('schedule', 'schedule', 'cosine', str)],
    'model': [
        ('c_s', 'singleFeatureDimension', 384, int), ('c_p', 'pairFeatureDimension', 128, int), ('rescale', 'rescale', 1, float),
        ('c_pos_emb', 'positionalEmbeddingDimension', 256, int), ('c_chain_emb', 'chainEmbeddingDimension', 64, int),
        ('c_timestep_emb', 'timestepEmbeddingDimension', 512, int),
        ('relpos_k', 'relativePositionK', 32, int), ('template_dist_min', 'templateDistanceMinimum', 2, float),
        ('template_dist_step', 'templateDistanceStep', 0.5, float), ('template_dist_n_bin', 'templateDistanceNumBins', 37, int),
        ('n_pair_transform_layer', 'numPairTransformLayers', 5, int),
        ('include_mul_update', 'includeTriangularMultiplicativeUpdate', True, None),
        ('include_tri_att', 'includeTriangularAttention', False, None),
        ('c_hidden_mul', 'triangularMultiplicativeHiddenDimension', 128, int),
        ('c_hidden_tri_att', 'triangularAttentionHiddenDimension', 32, int), ('n_head_tri', 'triangularAttentionNumHeads', 4, int),
        ('tri_dropout', 'triangularDropout', 0.25, float), ('pair_transition_n', 'pairTransitionN', 4, int),
        ('n_structure_layer', 'numStructureLayers', 8, int), ('n_structure_block', 'numStructureBlocks', 1, int),
        ('c_hidden_ipa', 'ipaHiddenDimension', 16, int), ('n_head_ipa', 'ipaNumHeads', 12, int),
        ('n_qk_point', 'ipaNumQkPoints', 4, int), ('n_v_point', 'ipaNumVPoints', 8, int), ('ipa_dropout', 'ipaDropout', 0.1, float),
        ('n_structure_transition_layer', 'numStructureTransitionLayers', 1, int),
        ('structure_transition_dropout', 'structureTransitionDropout', 0.1, float),
    ],
    'training': [
        ('seed', 'seed', 100, int), ('n_epoch', 'numEpoches', 1, int), ('batch_size', 'batchSize', 1, int),
        ('log_every_n_step', 'logEverySteps', 1000, int), ('checkpoint_every_n_epoch', 'checkpointEveryEpoches', 500, int),
        ('condition_loss_weight', 'conditionLossWeight', 1, int),
    ],
    'optimization': [('lr', 'learningRate', 1e-4, float)],
}


class Config:
    def __init__(self, filename=None):
        raw = {}
        if filename is not None:
            with open(filename) as fh:
                for line in fh:
                    parts = line.split()
                    if len(parts) == 2:
                        raw[parts[0]] = {'True': True, 'False': False}.get(parts[1], parts[1])
        for section, entries in _SPEC.items():
            table = {}
            for key, file_key, default, cast in entries:
                v = raw.get(file_key, default)
                table[key] = cast(v) if (cast is not None and v is not None) else v
            setattr(self, section, table)
