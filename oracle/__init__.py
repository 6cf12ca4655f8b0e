"""CPU ORACLE of PTT's hot path — TEST INFRASTRUCTURE ONLY.

index_ops  : C restatement (ptt_oracle.c) of FPS / ball query / gather / group / kNN.
             The four extension ops are "parity unpinned" (third-party, un-vendored,
             un-pinned CUDA dependency with no tests in the reference; see ptt_oracle.c).
dense_ref  : torch-CPU restatement of QueryAndGroup / SharedMLP / PointnetSAModuleVotes /
             PointNet2BackboneLight.branch_forward / TransformerBlock, pinned against the
             imported reference by tests/golden/make_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; ptt_amd/ never does.
"""
