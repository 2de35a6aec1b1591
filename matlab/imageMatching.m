function [allMatches, numMatches, tforms] = imageMatching(input, n, keypoints, matchesAll, imagesProcessed)
    %IMAGEMATCHING Shadows PP/imageMatching/imageMatching.m: the candidate pairs are chosen as the reference chooses
    %   them (:76-100), then ALL of them are verified in one device batch (aps_ransac_homography_batch) instead of one
    %   refineMatch call per parfor iteration (:121-156); outputs as the reference's (n x n cells, tforms{i,j} and its
    %   inverse in tforms{j,i}).
    %   input.useMATLABImageMatching = 1 (estgeotform2d) and input.showKeypointsPlot = 1 (the montage of refineMatch) are
    %   forwarded to the reference's own file.
    if input.useMATLABImageMatching == 1 || (isfield(input, 'showKeypointsPlot') && input.showKeypointsPlot == 1)
        [allMatches, numMatches, tforms] = aps_call_shadowed('imageMatching', mfilename('fullpath'), input, n, keypoints, matchesAll, imagesProcessed);
        return;
    end
    if numel(keypoints) ~= n
        error('imageMatching:InvalidKeypointsLength', 'keypoints must contain n elements (one per image).');
    end
    if numel(imagesProcessed) ~= n
        error('imageMatching:InvalidImagesLength', 'images must contain n elements (one per image).');
    end
    method = lower(input.imageMatchingMethod);
    if ~any(strcmp(method, {'ransac', 'mlesac'}))
        error('Valid image matching method is required.');
    end
    allMatches = cell(n);
    numMatches = zeros(n);
    tforms = cell(n, n);

    % candidate pairs: every image's m strongest partners by putative count, symmetrised, upper triangle (:76-100)
    cnt = cellfun(@(x) size(x, 1), matchesAll);
    sym = cnt + cnt.';
    sym(1:n + 1:end) = 0;
    [~, order] = sort(sym, 2, 'descend');
    keep = order(:, 1:min(input.mBrownLowe, n - 1));
    cand = false(n);
    rows = repmat((1:n).', 1, size(keep, 2));
    cand(sub2ind([n n], rows(:), keep(:))) = true;
    cand = triu(cand | cand.', 1);
    upIdx = find(cand);
    if isempty(upIdx), return; end
    [ri, ci] = ind2sub([n n], upIdx);
    fprintf('Image matching | Top-m filtering: %d pairs instead of %d pairwise image matches (%.1f%% reduction)\n', ...
        numel(upIdx), n * (n - 1) / 2, 100 * (1 - numel(upIdx) / (n * (n - 1) / 2)));

    % the pairs with at least four putative matches (:131-135) form the batch; their matched points one after the other
    nf = arrayfun(@(q) size(matchesAll{ri(q), ci(q)}, 1), (1:numel(upIdx)).');
    work = find(nf >= 4);
    if isempty(work), return; end
    pairPtr = [0; cumsum(nf(work))];
    p1 = zeros(pairPtr(end), 2);
    p2 = zeros(pairPtr(end), 2);
    for w = 1:numel(work)
        q = work(w);
        mt = matchesAll{ri(q), ci(q)};
        if max(mt(:, 1)) > size(keypoints{ri(q)}, 1) || max(mt(:, 2)) > size(keypoints{ci(q)}, 1)
            error('refineMatch:MatchIndexOutOfBounds', 'Match indices exceed keypoint array sizes.');
        end
        span = pairPtr(w) + 1:pairPtr(w + 1);
        p1(span, :) = keypoints{ri(q)}(mt(:, 1), :);
        p2(span, :) = keypoints{ci(q)}(mt(:, 2), :);
    end
    S = input.maxIter + 64;
    sampleIdx = aps_mex('ransac_draw_samples', nf(work), S, randi(2 ^ 31 - 1));
    % refineMatch estimates the map from image jj's points to image ii's (:242-245): matchedPts_2 first
    [models, mask, ~, ~] = aps_mex('ransac_homography_batch', p2, p1, pairPtr, input, sampleIdx, ...
        lower(input.transformationType), method);
    for w = 1:numel(work)
        q = work(w);
        inl = find(mask(pairPtr(w) + 1:pairPtr(w + 1)));
        if numel(inl) > 8 + 0.3 * nf(q)      % :150
            mt = matchesAll{ri(q), ci(q)};
            allMatches{ri(q), ci(q)} = mt(inl, :);
            numMatches(ri(q), ci(q)) = numel(inl);
            tforms{ri(q), ci(q)} = models(:, :, w);
            tforms{ci(q), ri(q)} = inv(models(:, :, w));
        end
    end
end
