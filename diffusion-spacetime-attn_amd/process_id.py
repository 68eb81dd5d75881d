# Key of the optional file side channel (reference: process_id.py:1). With prompt sharding each rank
# would need its own id; this build passes local-prompt embeddings in memory instead (sta.prompt_state).
NON_EXISTING_NAME_ID = 0
