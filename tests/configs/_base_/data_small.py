# base file of the test config: exercises `_base_` inheritance + recursive dict merge
data = dict(samples_per_gpu=2, workers_per_gpu=0, test=dict(type='TextMotionDataset', dataset_name='motionx'))
dist_params = dict(backend='nccl')
